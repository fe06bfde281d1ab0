#!/usr/bin/env python3
"""Benchmark of the whole-volume nuclei-detection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no ``WORLD_SIZE`` in the environment starts its N ranks itself (a child
``torch.distributed.run``, before this process touches the GPU) and relays rank 0's line.

Workloads (BASELINE.json configs; the default, c3, is the one the metric is quoted on):
  c3  2048 x 2048 x 1024 uint16, 5-sigma LoG scale space + NMS + prune   (configs[2]; N > 1: configs[3])
  c2  512 x 512 x 256 uint16, single-sigma LoG                            (configs[1])
  c5  2-channel 2048 x 2048 x 512 uint16 tile of a tiled light-sheet stack: per-block preprocessing
      (denoise_size 25), detection of both channels, intensity co-localisation, prune (configs[4])
At N > 1 the SAME volume's blocks are sharded over the ranks (strong scaling).  A *step* is one full pass of
the hot path over the volume: all blocks through the HIP kernels, the blob-table exchange, the overlap pruning;
the volume is resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel: algorithmic bytes (per VOLUME voxel) / HIP-event time on its
                launch stream against the 8 TB/s peak; the bytes it really moves and its VALU / MFMA busy fractions
                from the committed counter passes of this same code (null when the kernels changed since)
  cpu_baseline  the oracle (NumPy/SciPy restatement of the reference) on this box's host cores over a bounded
                sample of the same workload, detection and pruning seconds apart
  ranks         per-rank kernel / exchange / prune milliseconds (N > 1)
  sub_records   (default command on one GPU only) compact c2 and c5 results: value, ms_per_step, roofline
                fraction, parity of a sample against the oracle
"""
from __future__ import annotations

import argparse
import functools
import hashlib
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_BASE_PROFILE = dict(min_sigma_factor=3, max_sigma_factor=5, num_sigma=5, detection_threshold=0.1,
                     overlap=0.5, exclude_border=None, segment_size=256, denoise_size=None,
                     prune_tol_factor=(1, 1, 1), isotropic=None,
                     # preprocessing keys (only read with denoise_size): the reference's defaults
                     clip_vmin=5, clip_vmax=99.5, clip_min=0.2, clip_max=1.0, max_thresh_factor=0.5,
                     tot_var_denoise=None, unsharp_strength=0.3, erosion_threshold=0.2)
CONFIGS = {
    "c3": dict(shape=(1024, 2048, 2048), seed=3, channels=1, coloc=False, profile={},
               metric="Mvoxels/s, 2048x2048x1024 uint16 stack, 5-sigma LoG blob detection",
               what="sigma 3..5 x5"),
    "c2": dict(shape=(256, 512, 512), seed=2, channels=1, coloc=False,
               profile=dict(min_sigma_factor=3, max_sigma_factor=3, num_sigma=1),
               metric="Mvoxels/s, 512x512x256 uint16 stack, single-sigma LoG blob detection",
               what="sigma 3 x1"),
    "c5": dict(shape=(512, 2048, 2048), seed=3, channels=2, coloc=True, profile=dict(denoise_size=25),
               metric="Mvoxels/s (per channel pair), 2-channel 2048x2048x512 uint16 tile, preprocessing + "
                      "5-sigma LoG detection of both channels + intensity co-localisation",
               what="2 channels, denoise_size 25, sigma 3..5 x5, co-localisation"),
}
#: steps / warm-up of the compact sub-records the default command appends
SUB_RECORDS = {"c2": (100, 20), "c5": (8, 2)}       # (c2: ~1.3 ms a step -- 20 warm-up steps let the clocks and the host pool settle;
#                                                      c5: 0.25 s a step -- eight steps tell a 5 % change from noise, round 5's three did not)
RESOLUTIONS = np.array([[1.0, 1.0, 1.0]])
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
#: algorithmic HBM bytes per voxel per sigma of each kernel
ALG_BYTES = {"zpass": 2 + 8, "ypass": 8 + 8, "xpass": 8 + 4, "peaks": 4,
             # fused path (default): Z+X in one kernel (Gz / Gzz never leave the CU), then Y
             "zxpass": 2 + 8, "y2pass": 8 + 4}
B_ALG_PER_SIGMA = 50                # SURVEY.md section 8d contract figure
#: kernel families enqueued on the stream the LoG passes run on (everything but the side-stream table kernels)
MAIN_STREAM = ("zpass", "ypass", "xpass", "generic", "peaks", "zxpass", "y2pass", "preproc", "zxpack", "rescore")
#: committed counter passes of this command (tools/profile_round.sh): HBM bytes per launch, VALU / MFMA busy
PMC_FILE = "profiles/r06_pmc_counters.json"
ZX_DTYPES = {7: "f16x2 MFMA (f32 accumulate) + 16-bit fixed-point intermediates; f64 re-score of every candidate",
             6: "f16x2 MFMA (f32 accumulate), f32 intermediates; f64 re-score of every candidate",
             2: "f32 (packed VALU); f64 re-score of every candidate",
             0: "f32 (separate passes); f64 re-score of every candidate"}


# ------------------------------------------------------------------ CPU baseline (oracle)
def _cpu_block(args):
    """One block through the oracle (runs in a spawned worker: NumPy/SciPy only)."""
    coord, offset, last_coord, sub, profiles, dms, channel, coloc = args
    near_max = [-1.0] * max(1, len(profiles))
    from oracle import magmap_oracle as mmo
    return coord, mmo.detect_sub_roi(coord, offset, last_coord, None, sub, channel, profiles,
                                     RESOLUTIONS, denoise_max_shape=dms, near_max=near_max, coloc=coloc)


def cpu_baseline(sample: np.ndarray, cores: int, profile: dict, channels, coloc: bool, with_colocs: bool = False):
    """Reference strategy (magmap/cv/stack_detect.py:222-257): a process pool over blocks.  ``with_colocs``: the
    first value is ``(final table, co-localisation flags)`` as ``detect_blobs_blocks`` leaves them (:463-467)."""
    from oracle import magmap_oracle as mmo
    t0 = time.time()
    blocks = mmo.setup_blocks(profile, sample.shape[:3], RESOLUTIONS)
    sl, off = blocks["sub_roi_slices"], blocks["sub_rois_offsets"]
    last = np.subtract(sl.shape, 1)
    profiles = [profile] * max(1, len(channels))
    chl = list(channels) if sample.ndim > 3 else None
    jobs = [(c, off[c], last, sample[sl[c]], profiles, blocks["denoise_max_shape"], chl, coloc)
            for c in np.ndindex(*sl.shape)]
    seg = np.zeros(sl.shape, dtype=object)
    with mp.get_context("spawn").Pool(processes=min(cores, len(jobs))) as pool:
        for coord, tbl in pool.imap_unordered(_cpu_block, jobs):
            seg[coord] = tbl
    t_detect = time.time() - t0
    pruned, _ = mmo.prune_blobs_mp(sample.shape[:3], seg, blocks["overlap"], blocks["tol"], sl, off,
                                   list(channels), blocks["overlap_padding"])
    final = colocs = None
    if pruned is not None:
        pruned[:, 0:3] = pruned[:, 7:10]
        if coloc:
            colocs = pruned[:, 10:10 + sample.shape[3]].astype(np.uint8)
        final = pruned[:, [0, 1, 2, 3, 4, 5, 6, 10]]
    return ((final, colocs) if with_colocs else final), t_detect, time.time() - t0, len(jobs)


def canon(t):
    return t[np.lexsort(tuple(t[:, i] for i in range(t.shape[1] - 1, -1, -1)))]


def physical_cores() -> int:
    """Distinct (package, core) pairs of /proc/cpuinfo; os.cpu_count() when that is not readable."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def source_digest() -> str:
    """SHA-1 over the DEVICE sources the library is built from (kernels, their headers, the Makefile's flags; not
    the host-only table code ``mmx_host.cpp``, which no counter of a kernel depends on): the committed counter
    passes name the digest they were taken with, and a line only quotes them while the sources are still those."""
    h = hashlib.sha1()
    src = os.path.join(ROOT, "magellanmapper_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if (name.endswith((".hip", ".inc", ".h", ".cpp")) and name != "mmx_host.cpp") or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()


def self_launch(argv, n: int) -> int:
    """Start the ``n`` ranks as a child ``torch.distributed.run`` and relay what rank 0 prints.  Runs before this
    process has imported torch or touched the GPU (a process that has initialised the GPU must not be replaced, and
    the children must find the GPUs unclaimed)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def make_host_sample(shp, seed, n_chl):
    """Host sample of the workload from the same generator (per-channel seeds, channel 1 shares 70 % of channel 0)."""
    import torch
    from magellanmapper_amd import synth as _synth
    cpu_dev = torch.device("cpu")
    c0 = _synth.make_volume_device(shp, seed, cpu_dev).to(torch.int32)
    if n_chl == 1:
        return c0.numpy().astype(np.uint16)
    c1 = _synth.make_volume_device(shp, seed + 1, cpu_dev).to(torch.int32)
    c1 = torch.maximum(c1, (c0 * 7) // 10)
    return torch.stack((c0, c1), dim=-1).numpy().astype(np.uint16)


def config_setup(name: str, args, host_vol):
    """``(cfg, profile, shape, n_channels, coloc)`` of a named workload with the command line's overrides."""
    cfg = CONFIGS[name]
    profile = dict(_BASE_PROFILE, **cfg["profile"])
    if args.denoise and name == args.config:
        profile["denoise_size"] = args.denoise
    if args.segment_size and name == args.config:
        profile["segment_size"] = args.segment_size
    use_vol = host_vol if name == args.config else None
    shape = tuple(use_vol.shape[:3]) if use_vol is not None else (
        tuple(args.shape) if (args.shape and name == args.config) else cfg["shape"])
    n_chl = (use_vol.shape[3] if use_vol.ndim > 3 else 1) if use_vol is not None else cfg["channels"]
    return cfg, profile, shape, n_chl, bool(cfg["coloc"] and n_chl > 1)


def run_cpu_baseline(name, args, host_vol):
    """The oracle on a bounded sample of workload ``name`` (before the GPU is initialised: the worker pool is spawned)."""
    cfg, profile, shape, n_chl, coloc = config_setup(name, args, host_vol)
    phys = physical_cores()
    cores = args.cpu_cores or phys
    pool_note = None
    if n_chl > 1 and not args.cpu_cores and not (args.cpu_full and name == args.config) and cores > 16:
        # the preprocessed two-channel sample: the pool's throughput on this pool's 128-core boxes is FLAT from 16
        # processes on (tools/exp/cpu_sample_scaling.py: 2.74 / 2.42 / 2.57 Mvoxel/s with 16 / 32 / 64 blocks and
        # processes -- detection 37 / 94 / 157 s), so the larger sample only made the default command 2.4 minutes longer
        cores = 16
        pool_note = ("16 blocks over 16 processes: the pool's throughput on this box does not grow beyond that "
                     "(measured flat for 16 / 32 / 64 processes with one block each, profiles/r06_experiments.txt section 10)")
    # a bounded sample (10-30 s of CPU work): about one block per core, at least two z-layers of blocks where
    # the volume has them so that the sample prunes seams along all three axes
    want_blocks = max(2, cores)
    # (c5: both channels are preprocessed tile by tile in Python loops and detected -- one layer of 96-plane
    #  blocks keeps the oracle at tens of seconds)
    bz = 2 if (shape[0] > 256 and n_chl == 1) else 1
    by = max(1, min(shape[1] // 256, int(np.sqrt(want_blocks / bz) + 0.5)))
    bx = max(1, min(shape[2] // 256, -(-want_blocks // (bz * by))))
    sz = min(shape[0], 320 if bz == 2 else (96 if n_chl > 1 else 256))
    sshape = (sz, min(shape[1], 256 * by), min(shape[2], 256 * bx))
    if args.cpu_full and name == args.config:
        sshape = tuple(shape)               # the SAME volume (minutes of CPU work: not the default)
    use_vol = host_vol if name == args.config else None
    sample = make_host_sample(sshape, cfg["seed"], n_chl) if use_vol is None else np.ascontiguousarray(
        use_vol[:sshape[0], :sshape[1], :sshape[2]])
    channels = list(range(n_chl))
    (cpu_final, cpu_colocs), t_det, t_tot, n_jobs = cpu_baseline(sample, cores, profile, channels, coloc, with_colocs=True)
    cpu = {"value": round(int(np.prod(sshape)) / t_tot / 1e6, 3), "unit": "Mvoxels/s",
           "cores": min(cores, n_jobs), "kind": "port",
           "cpu_count": os.cpu_count(), "physical_cores": phys,
           "detection_s": round(t_det, 2), "pruning_s": round(t_tot - t_det, 2),
           "whole_volume": tuple(sshape) == tuple(shape),
           "sample": f"{sshape[0]}x{sshape[1]}x{sshape[2]} (z,y,x){' x %d channels' % n_chl if n_chl > 1 else ''} "
                     f"volume from the same generator (seed, blob density, profile, segment_size as the GPU run), "
                     f"{n_jobs} blocks over a pool of {min(cores, n_jobs)} processes "
                     f"(reference strategy, stack_detect.py:222-257)",
           "blobs": 0 if cpu_final is None else int(len(cpu_final))}
    if pool_note:
        cpu["pool_note"] = pool_note
    return dict(cpu=cpu, final=cpu_final, sample=sample, colocs=cpu_colocs)


def volume_sha1(vol: np.ndarray) -> str:
    return hashlib.sha1(np.ascontiguousarray(vol).tobytes()).hexdigest()


def load_parity_sample(path, name, args):
    """A committed oracle table instead of the oracle itself (``--parity-sample``; made by
    ``tests/golden/make_bench_samples.py``, which runs :func:`cpu_baseline` on a fixed sample of the workload): the
    sample volume is generated again here and must hash to what the table was made from -- otherwise ``None`` and the
    caller runs the oracle pool as usual.  Gives the parity check without the minutes of CPU work, no CPU timing."""
    g = np.load(path, allow_pickle=False)
    cfg, profile, _, n_chl, coloc = config_setup(name, args, None)
    if str(g["config"]) != name or int(g["n_channels"]) != n_chl or json.loads(str(g["profile"])) != json.loads(
            json.dumps(profile)):
        raise SystemExit(f"--parity-sample {path}: made for another workload / profile than {name}")
    sample = make_host_sample(tuple(int(v) for v in g["shape"]), int(g["seed"]), n_chl)
    if volume_sha1(sample) != str(g["volume_sha1"]):
        print(f"bench.py: the sample volume generated here differs from the one {path} was made from "
              "(another torch build / CPU?): running the oracle instead", file=sys.stderr)
        return None
    final = np.asarray(g["final"], dtype=np.float64)
    return dict(cpu=None, final=final if len(final) else None, sample=sample,
                colocs=np.asarray(g["colocs"]) if "colocs" in g.files else None, committed=os.path.basename(path))


def run_gpu(name, args, host_vol, baseline, steps, warmup, ctx):
    """One workload on the GPU(s): warm-up, the timed region, the record (rank 0; ``None`` elsewhere)."""
    import torch
    import torch.distributed as tdist
    from magellanmapper_amd import _native as nat
    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import config, detector, dist, stack_detect, synth
    rank, world, dev, backend = ctx["rank"], ctx["world"], ctx["dev"], ctx["backend"]
    cfg, PROFILE, shape, n_chl, coloc = config_setup(name, args, host_vol)
    use_vol = host_vol if name == args.config else None
    channels = list(range(n_chl))
    seed = cfg["seed"]

    config.resolutions = RESOLUTIONS
    config.filename = "bench"
    config.setup_roi_profiles(None)
    config.roi_profile.update(PROFILE)
    for p in config.roi_profiles:
        p.update(PROFILE)
    config.near_max = [-1.0] * max(1, n_chl)
    blocks = stack_detect.setup_blocks(config.roi_profile, shape)
    grid = blocks.sub_roi_slices.shape
    n_blocks = int(np.prod(grid))

    # Each rank needs the z-range its blocks touch (blocks are z-major contiguous per rank).
    coords = list(np.ndindex(*grid))
    # --share k/N: this ONE process stands in for rank k of N (dist.Loopback: a recording in place of the wire)
    share, wire = ctx.get("share"), ctx.get("wire")
    eff_rank, eff_world = share if share else (rank, world)
    lo, hi = dist.share_bounds(n_blocks, eff_rank, eff_world)
    zs = [blocks.sub_roi_slices[coords[i]][0] for i in range(lo, hi)]
    z0 = min(s.indices(shape[0])[0] for s in zs) if zs else 0
    z1 = max(s.indices(shape[0])[1] for s in zs) if zs else 1
    if share:
        z0, z1 = 0, shape[0]            # (the whole volume: the recording rounds play every rank once)
    t_gen = time.time()
    if use_vol is not None:
        slab = torch.from_numpy(np.ascontiguousarray(use_vol[z0:z1]).view(np.int16)).to(dev).view(torch.uint16)
    else:
        slab = synth.make_volume_device(shape, seed, dev, z_range=(z0, z1))
        if n_chl > 1:     # channel 1: its own blob field plus 70 % of channel 0's (co-localised blobs)
            c1 = synth.make_volume_device(shape, seed + 1, dev, z_range=(z0, z1))
            c1 = torch.maximum(c1.to(torch.int32), (slab.to(torch.int32) * 7) // 10).to(slab.dtype)
            slab = torch.stack((slab, c1), dim=-1).contiguous()
            del c1
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    # A rank-local view that behaves like the full (z, y, x[, c]) ROI for block addressing.
    # (DeviceVolume(z_off=, full_shape=): the planes a rank holds, addressed by their coordinates in the whole volume)
    class SlabVolume(bl.DeviceVolume):
        def __init__(self, t, z_off, full_shape):
            # (host sources -- `--from-host` -- go up beside the detection: this script leaves them alone meanwhile)
            super().__init__(t, dev, streamed=True,
                             cells=None if args.upload_order == "slabs" else stack_detect._upload_cells(blocks.sub_roi_slices, full_shape),
                             z_off=z_off, full_shape=tuple(full_shape)[:3])

    dvol = SlabVolume(slab, z0, shape)
    timers = {"gather_ms": 0.0, "prune_ms": 0.0, "detect_ms": 0.0, "tail_ms": 0.0, "tail_exchange_ms": 0.0,
              "start_ms": 0.0, "last_batch_host_ms": 0.0}

    def finish(pruned):
        if pruned is None:
            return None, None
        if isinstance(pruned, stack_detect._FinalTable):       # the pruning step wrote the final columns itself
            detector.Blobs(None).cols = list(pruned.col_names)
            return pruned.view(np.ndarray), (None if pruned.coloc_cols is None else pruned.coloc_cols.astype(np.uint8))
        bb = detector.Blobs(pruned)              # the table's final form (reference stack_detect.py:458-467)
        bb.replace_rel_with_abs_blob_coords(pruned)
        colocs = pruned[:, 10:10 + n_chl].astype(np.uint8) if coloc else None
        return bb.remove_abs_blob_coords(True), colocs

    def detect_and_prune(vol, blk):
        """What ``stack_detect.detect_blobs_blocks`` does between loading the image and writing its files: every rank
        detects its share of the blocks; the pruning then is either a collective of all ranks (each prunes its own
        rows) or, where the block geometry does not allow that, rank 0's after a gather."""
        t_a = time.perf_counter()
        stack_detect.StackDetector.plan_pruning(blk.overlap, blk.tol, blk.overlap_padding, channels)
        seg = stack_detect.StackDetector.detect_blobs_sub_rois(
            None, vol, blk.sub_roi_slices, blk.sub_rois_offsets, blk.denoise_max_shape,
            blk.exclude_border, coloc, channels)
        st = stack_detect.StackDetector.last_stats
        t_b = time.perf_counter()
        final = colocs = None
        if rank == 0 or getattr(seg, "local_only", False):
            # (final_form, untouched: as stack_detect._StackRun calls it -- no co-localisation columns, one process or
            #  every rank pruning its own rows; the tables come straight from the call above)
            pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
                vol, seg, blk.overlap, blk.tol, blk.sub_roi_slices, blk.sub_rois_offsets,
                channels, blk.overlap_padding, untouched=True,
                final_form=dist.world_size() == 1 or getattr(seg, "local_only", False),
                n_flag_cols=n_chl if coloc else 0)
            if rank == 0:
                final, colocs = finish(pruned)
        t_c = time.perf_counter()
        # (time inside the collectives -- the table gather, or the two exchanges of the distributed pruning -- apart
        #  from this rank's own detection and pruning work)
        exchange = dist.last_gather_ms()
        in_prune = exchange if getattr(seg, "local_only", False) else 0.0
        timers["gather_ms"] += exchange
        timers["detect_ms"] += (t_b - t_a) * 1e3 - (exchange - in_prune)
        timers["prune_ms"] += (t_c - t_b) * 1e3 - in_prune
        # what this rank still did after its last kernel had finished: the last batch's host work, the pruning (its
        # exchanges included) and the final columns
        timers["tail_ms"] += (t_c - bl.LAST_BATCH_DONE_T) * 1e3 if bl.LAST_BATCH_DONE_T >= t_a else 0.0
        timers["tail_exchange_ms"] += in_prune
        # step begin -> first batch handed to the GPU; last batch seen done -> detect_blobs_sub_rois returns
        timers["start_ms"] += (bl.FIRST_ENQUEUED_T - t_a) * 1e3 if bl.FIRST_ENQUEUED_T >= t_a else 0.0
        timers["last_batch_host_ms"] += (t_b - bl.LAST_BATCH_DONE_T) * 1e3 if bl.LAST_BATCH_DONE_T >= t_a else 0.0
        return final, colocs, st

    def one_step():
        if wire is not None:
            wire.begin_step(eff_rank)
        return detect_and_prune(dvol, blocks)

    # workspace budget per batch: from the free HBM of this GPU unless given (ranks that share a GPU in the
    # functional tests share its memory)
    if args.budget_gb > 0:
        budget = int(args.budget_gb * (1 << 30))
    else:
        free_b, _ = torch.cuda.mem_get_info()
        sharers = max(1, -(-world // max(1, torch.cuda.device_count()))) if backend != "nccl" else 1
        budget = min(16 << 30, int(0.55 * free_b / sharers))
    bl.blob_log_blocks = functools.partial(ctx["blob_log_blocks"], budget_bytes=budget)
    bl.BUDGET_BYTES = budget               # (calls that do not name a budget: the multi-channel pipeline)

    # ---------------- the HIP path on the CPU-baseline sample must give the identical table
    parity = None
    cpu = None
    if baseline is not None:
        # (several ranks: rank 0 holds the oracle's table and runs the sample on its own GPU as one process would --
        #  dist.solo() -- while the others wait at the barrier below)
        cpu, cpu_final, sample = baseline["cpu"], baseline["final"], baseline["sample"]
        sblocks = stack_detect.setup_blocks(config.roi_profile, sample.shape[:3])
        sdvol = bl.DeviceVolume(sample)
        with dist.solo():
            gpu_final, gpu_colocs, _ = detect_and_prune(sdvol, sblocks)
        parity = bool(cpu_final is not None and gpu_final is not None and gpu_final.shape == cpu_final.shape and
                      np.array_equal(canon(gpu_final), canon(cpu_final)))
        if parity and baseline.get("colocs") is not None:
            # (same row order as the oracle's table: compared above in canonical order, here row for row)
            parity = bool(np.array_equal(gpu_final, cpu_final) and gpu_colocs is not None and
                          np.array_equal(gpu_colocs, baseline["colocs"]))
        del sdvol

    # ---------------- warm-up, then the timed region
    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    if wire is not None:
        # the recording: every rank of the N played twice by this process (the first round puts the rows near the seams
        # on the tape, the second the survivors pruned with them), then rank k's steps replay it
        wire.mode = "record"
        for _ in range(2):
            for q in range(eff_world):
                wire.begin_step(q)
                detect_and_prune(dvol, blocks)
        wire.mode = "replay"
    nat.timing_enable(True)             # (the warm-up tells which kernel family dominates: the one the roofline is about)
    # Volumes of a few blocks (c2) run their timed region as replays of a captured hipGraph (below), which the per-kernel
    # timing prevents: their last two warm-up steps run without it -- the first sighting of the batch and its capture,
    # with the stream the graph runs on taking its hardware queue (12-17 ms once) -- so that the timed region is what a
    # caller's steady state is: replays.
    early_replay = (world == 1 and 0 < n_blocks <= bl.GRAPH_BLOCKS and bl.NATIVE_BATCH and not PROFILE["denoise_size"]
                    and warmup >= 3)
    warm = {}
    for w in range(warmup):
        if early_replay and w == warmup - 2:
            torch.cuda.synchronize()
            warm = {k: ms for k, (ms, n) in nat.timing_read().items() if n and k in ALG_BYTES}
            nat.timing_enable(False)
        one_step()
    torch.cuda.synchronize()
    if not early_replay:
        warm = {k: ms for k, (ms, n) in nat.timing_read().items() if n and k in ALG_BYTES}
    nat.timing_enable(False)
    for k in timers:
        timers[k] = 0.0
    # Volumes of a few blocks (c2): their one batch is replayed as a captured hipGraph (blob_log.GRAPH_BLOCKS), which the
    # per-kernel event timing would prevent -- a capture cannot hold its events.  The timed region then runs WITHOUT the
    # per-kernel timing, as a caller's step does, and the per-kernel times come from extra steps after it.
    replayed = world == 1 and 0 < n_blocks <= bl.GRAPH_BLOCKS and bl.NATIVE_BATCH and not PROFILE["denoise_size"]
    # Larger volumes: every kernel family records its HIP events inside the timed region (--kernel-events all, the
    # default; `roofline` is the dominant family's live measurement over that region).  What those ~400 events per step
    # cost was measured with --kernel-events dominant | none (only the dominant family's events inside the region |
    # none, the other families timed in extra steps after it): 102.2 / 102.8 ms with all, 103.3 / 102.7 with the
    # dominant family's, 102.2 / 102.1 with none, alternating on one box -- nothing beyond the run-to-run spread.
    events = "none" if replayed else args.kernel_events
    if events == "dominant" and not warm:
        events = "all"                  # (no warm-up step to tell the dominant family from)
    dominant = max(warm, key=warm.get) if events == "dominant" else None
    nat.timing_enable(events != "none", kinds=None if events != "dominant" else [dominant])
    bl.PRE_WAITS.clear()
    replays0 = bl.GRAPH_REPLAYS
    barrier()
    t0 = time.perf_counter()
    final = colocs = None
    stats = None
    step_ends = []
    for _ in range(steps):
        final, colocs, stats = one_step()
        step_ends.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    each = np.diff(np.concatenate(([t0], step_ends))) * 1e3
    step_timers = dict(timers)          # (the timed region's: the extra steps below keep adding to `timers`)
    n_replays = bl.GRAPH_REPLAYS - replays0
    pre_wait_ms = sum(a.elapsed_time(b) for a, b in bl.PRE_WAITS) / steps if bl.PRE_WAITS else None
    bl.PRE_WAITS.clear()
    if events != "all":
        timed = nat.timing_read() if events == "dominant" else {}
        extra = max(1, min(steps, 10))
        nat.timing_enable(True)
        for _ in range(extra):
            one_step()
        torch.cuda.synchronize()
        ktimes = {k: (ms * steps / extra, n * steps // extra) for k, (ms, n) in nat.timing_read().items()}
        if dominant is not None and timed.get(dominant, (0.0, 0))[1]:
            ktimes[dominant] = timed[dominant]          # (the timed region's own events)
    else:
        ktimes = nat.timing_read()
    nat.timing_enable(False)
    per_rank = None
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([sum(ms for ms, n in ktimes.values()) / steps, step_timers["detect_ms"] / steps,
                             step_timers["gather_ms"] / steps, step_timers["prune_ms"] / steps, float(hi - lo),
                             step_timers["tail_ms"] / steps,
                             (step_timers["tail_ms"] - step_timers["tail_exchange_ms"]) / steps],
                            dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        tdist.all_gather(allr, mine)
        per_rank = [dict(zip(("kernel_ms", "detect_wall_ms", "gather_ms", "prune_ms", "blocks",
                              "tail_after_last_kernel_ms", "tail_without_exchanges_ms"),
                             (round(float(v), 2) for v in r.cpu()))) for r in allr]
    host_run = None
    if args.from_host and world == 1:
        host_run = lambda rec: from_host_record(args.from_host, host_src, z0, shape, SlabVolume, detect_and_prune,
                                                blocks, rec, min(steps, 5), args.tiles)
        host_src = torch.empty(tuple(slab.shape), dtype=slab.dtype).pin_memory()
        host_src.copy_(slab)
        torch.cuda.synchronize()
    del dvol, slab
    bl.release_buffers()
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    nvox = int(np.prod(shape))
    ns = PROFILE["num_sigma"]
    # VOLUME voxels (all channels) of this rank's share: the contract's per-voxel figures count volume voxels, the
    # overlap between blocks (x 1.05 at the benchmark geometry) is the builder's loss
    my_vox = nvox * n_chl * (hi - lo) / max(1, n_blocks)
    flags = []
    per_kernel = {}
    for k, (ms, n) in ktimes.items():
        if n:
            per_kernel[k] = {"ms_per_step": round(ms / steps, 3), "launches_per_step": n // steps}
            if k in ALG_BYTES:
                gbs = ALG_BYTES[k] * my_vox * ns * steps / (ms * 1e-3) / 1e9
                per_kernel[k]["alg_GBps"] = round(gbs, 1)
                if gbs > HBM_PEAK_GBS:
                    flags.append(f"kernels.{k}.alg_GBps")
                    per_kernel[k]["note"] = (
                        "above the 8 TB/s peak: quoted on the contract's byte count (peaks: 4 B per voxel and sigma), while "
                        "the kernel reads the sparse NMS entries the Y pass leaves (16 B per 64 voxels and sigma plus the "
                        "lines of the set bits)" if k == "peaks" else
                        "above the 8 TB/s peak: quoted on the contract's byte count (Y pass: 8 B in + 4 B out per voxel and "
                        "sigma), while the kernel reads 16-bit tiles (4.35 B per voxel) and stores the LoG only where "
                        "something is above the threshold (~1.2 B per voxel): see roofline.traffic / DESIGN.md section 4"
                        if k == "y2pass" else "above the 8 TB/s peak on the contract's byte count: the kernel moves fewer bytes")
            elif k == "preproc":      # once per voxel (not per sigma): 2 B in, 8 + 4 B out; fp64-VALU bound
                per_kernel[k]["alg_GBps"] = round(14 * my_vox * steps / (ms * 1e-3) / 1e9, 1)
                # ~220 float64 operations per voxel (three sigma-8 passes of 68 + stretch, unsharp, erosion), none
                # fused (SciPy's arithmetic has no FMA), against 256 CUs x 64 lanes x 2.4 GHz = 39.3 Top/s
                top = 220.0 * my_vox * steps / (ms * 1e-3) / 1e12
                per_kernel[k]["f64_Tops"] = round(top, 2)
                per_kernel[k]["frac_of_f64_vector_rate_no_fma"] = round(top / 39.3, 3)
    stream_k = {k: v for k, v in per_kernel.items() if k in ALG_BYTES}
    dom = max(stream_k, key=lambda k: stream_k[k]["ms_per_step"]) if stream_k else None
    zx_path = bl.LAST_ZX_PATH
    roof = None
    if dom:
        launches = ktimes[dom][1]
        avg_ms = ktimes[dom][0] / max(1, launches)
        # what the committed counter passes of this command say about the same kernel, while the sources they were
        # taken with are still the ones this library was built from
        traffic = actual_frac = valu_busy = mfma_busy = None
        counters_note = None
        try:
            with open(os.path.join(ROOT, PMC_FILE)) as f:
                pmc = json.load(f)
            fresh = pmc.get("source_digest") == source_digest()
            # (the counter passes ran the DEFAULT command: same volume, raw voxels, default batches, one rank)
            usable = (fresh and name == "c3" and tuple(shape) == cfg["shape"] and world == 1 and not share and
                      not PROFILE["denoise_size"] and use_vol is None and not args.budget_gb and
                      not args.segment_size and dom in pmc.get("per_launch_GB", {}))
            if usable:
                traffic = round(pmc["per_launch_GB"][dom]["total_GB"] * 1e9)
                actual_frac = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                busy = pmc.get("busy", {}).get(dom, {})
                valu_busy, mfma_busy = busy.get("valu_busy"), busy.get("mfma_busy")
                counters_note = (PMC_FILE + ": rocprofv3 --pmc passes of this command (separate FETCH_SIZE / WRITE_SIZE "
                                 "/ SQ passes, calibrated as MI355X_MICROARCH.md prescribes); bytes per launch averaged "
                                 "over all launches of the kernel; not collected in this run")
            elif not fresh:
                counters_note = PMC_FILE + " was taken with other kernel sources (digest mismatch): not quoted"
        except (OSError, KeyError, ValueError):
            counters_note = "no committed counter passes for this code"
        roof = {"bound": "hbm", "kernel": dom, "achieved": stream_k[dom]["alg_GBps"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(stream_k[dom]["alg_GBps"] / HBM_PEAK_GBS, 4),
                "alg_bytes_per_launch": int(ALG_BYTES[dom] * my_vox * ns * steps / max(1, launches)),
                "avg_launch_ms": round(avg_ms, 4),
                "traffic": traffic, "actual_frac": actual_frac, "valu_busy": valu_busy, "mfma_busy": mfma_busy,
                "counters": counters_note,
                "copy_GBps": ctx.get("copy_gbps"),
                "frac_of_copy": None if not ctx.get("copy_gbps") else round(stream_k[dom]["alg_GBps"] / ctx["copy_gbps"], 4),
                "note": "achieved = SURVEY.md 8d's algorithmic bytes of this pass (zxpass: 2 B voxels in + 8 B "
                        "intermediates out per volume voxel and sigma) / its HIP-event time; the kernel itself moves "
                        "fewer bytes than that ('traffic'): its intermediates are 16-bit fixed point, and with the VALU and "
                        "the MFMA pipe each about half busy it waits for its L2 requests, not for HBM (DESIGN.md section 4b)"
                        + ("; its HIP-event time includes sharing the GPU with the NMS and re-score kernels of the previous "
                           "batch on the second stream (alone: ~10 % less, MMX_RESCORE_STREAM=0)"
                           if (bl.RESCORE_STREAM and not PROFILE["denoise_size"]) else "")
                        if dom == "zxpass" else None}
        if dom == "zxpass" and zx_path in (nat.MMX_ZX_TILED, nat.MMX_ZX_TILED_Q16):
            # the same kernel against the matrix-core roofline: MFMAs it issues (16 x 16 x 32 float16, 16 384 flop
            # each) per 16 x 16 tile step -- X pass 12 (16-bit tiles) or 16 per two k-steps, 6 / 8 for radius <= 8;
            # Z pass 9 per k-step -- over every block row, column tile and z step of this rank's blocks
            space = bl.ScaleSpace.make(PROFILE["min_sigma_factor"] * detector.calc_scaling_factor()[2],
                                       PROFILE["max_sigma_factor"] * detector.calc_scaling_factor()[2], ns)
            q16 = zx_path == nat.MMX_ZX_TILED_Q16
            flop = 0.0
            for i in range(lo, hi):
                shp = [s_.indices(n_)[1] - s_.indices(n_)[0] for s_, n_ in zip(blocks.sub_roi_slices[coords[i]], shape)]
                for R in space.radii:
                    nkx, la = (1, 1) if R <= 8 else ((2, 1) if R <= 16 else (2, 2))
                    per_step = nkx * (6 if q16 else 8) + (la + 1) * 9
                    flop += shp[1] * -(-shp[2] // 16) * (-(-shp[0] // 16) + la) * per_step * 16384.0
            tfs = flop * n_chl * steps / (ktimes[dom][0] * 1e-3) / 1e12
            roof["mfma"] = {"achieved_TFLOPs": round(tfs, 1), "peak_TFLOPs": 2500.0, "frac": round(tfs / 2500.0, 4),
                            "note": "float16 MFMA flops the kernel issues (split-float16 products: 3 MFMAs per float32 "
                                    "product) over its duration, against the dense float16 peak"}
    gpu_ms = sum(ms for ms, n in ktimes.values()) / steps
    # (per-block preprocessing runs on a stream of its own beside the LoG kernels: its spans overlap theirs)
    overlapped = ("preproc",) if (PROFILE["denoise_size"] and bl.PRE_STREAM) else (
        ("peaks", "rescore") if (bl.RESCORE_STREAM and not PROFILE["denoise_size"]) else ())
    main_ms = sum(ms for k, (ms, n) in ktimes.items() if k in MAIN_STREAM and k not in overlapped) / steps
    b_alg = B_ALG_PER_SIGMA * ns * n_chl
    vol_bytes = nvox * n_chl * 2
    frac_kernels = b_alg * (nvox / eff_world) / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    frac_wall = b_alg * nvox / (elapsed / steps) / 1e9 / HBM_PEAK_GBS / eff_world
    for key, val in (("pipeline_roofline.frac_kernels", frac_kernels), ("pipeline_roofline.frac_wall", frac_wall),
                     ("roofline.frac", roof["frac"] if roof else 0)):
        if val > 1:
            flags.append(key)
    h2d_gbps = ctx.get("h2d_gbps")
    out = {
        "metric": cfg["metric"],
        "value": round(nvox * steps / elapsed / 1e6, 2), "unit": "Mvoxels/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 2), "higher_is_better": True,
        # host clock after every step of the timed region (the last step's barrier is not in these): spread of a step
        "step_ms_percentiles": {"min": round(float(each.min()), 3), "p50": round(float(np.percentile(each, 50)), 3),
                                "p95": round(float(np.percentile(each, 95)), 3), "max": round(float(each.max()), 3)},
        "scaling": "strong", "vs_baseline": None,
        "dtype": ZX_DTYPES.get(zx_path, "f32; f64 re-score of every candidate") if PROFILE["denoise_size"] is None
                 else "f64 preprocessing; " + ZX_DTYPES.get(zx_path, "f32; f64 re-score of every candidate"),
        "zx_path": zx_path, "host_path": bl.HOST_PATH,
        "kernel_events": {"in_timed_region": events if events != "dominant" else [dominant],
                          "note": "kernel families whose HIP events were recorded inside the timed region (--kernel-events; "
                                  "families not timed there are timed in extra steps after it)"},
        "graph_replay": (n_replays if replayed else None),
        "graph_replay_note": None if not replayed else (
            "graph_replay = batches of the timed region that were replayed from a captured hipGraph (no per-kernel events inside "
            "it); `kernels` and `roofline` come from extra steps run after it with the launches made one by one"),
        "y_kernel": (None if zx_path != nat.MMX_ZX_TILED_Q16 else
                     "y6_kernel (VALU taps)" if bl.ZX_FLAGS & nat.MMX_ZX_Y_VALU else "ym_kernel (matrix cores)"),
        "data": "synthetic" if use_vol is None else args.volume,
        "config": {"workload": f"{name}: {shape[2]}x{shape[1]}x{shape[0]} (x,y,z) uint16 Gaussian-blob volume, "
                               f"seed {seed}, {n_blocks} blocks (segment_size {PROFILE['segment_size']}, overlap 5), {cfg['what']}, "
                               "threshold 0.1, overlap 0.5; detect + table exchange + prune"
                               + (f"; per-block preprocessing (denoise_size {PROFILE['denoise_size']}: saturate + "
                                  "unsharp + erosion in float64)" if PROFILE["denoise_size"] else ""),
                   "blocks_per_rank": hi - lo, "parallelism": f"blocks sharded over {world} GPU(s)",
                   "batch_budget_GB": round(budget / (1 << 30), 1)},
        "blobs": 0 if final is None else int(len(final)),
        # digest of the final 8-column table of the last step (computed after the timed region): the same
        # for every path / batch size / rank count that is correct (tests compare smaller volumes with the
        # oracle row by row; this is the full-size cross-check)
        "table_sha1": None if final is None else hashlib.sha1(np.ascontiguousarray(final).tobytes()).hexdigest(),
        "blobs_per_s": round((0 if final is None else len(final)) * steps / elapsed, 1),
        "roofline": roof,
        "pipeline_roofline": {
            "alg_bytes_per_voxel": b_alg,
            "gpu_kernel_ms_per_step_rank0": round(gpu_ms, 2),
            "main_stream_kernel_ms_per_step_rank0": round(main_ms, 2),
            "achieved_GBps_kernels": round(frac_kernels * HBM_PEAK_GBS, 1),
            "frac_kernels": round(frac_kernels, 4),
            "achieved_GBps_wall": round(frac_wall * HBM_PEAK_GBS * world, 1),
            "frac_wall": round(frac_wall, 4),
            # wall clock not covered by a kernel on the stream the LoG passes run on: start-up before the first
            # launch, launch gaps, waits for the preprocessing stream, and the tail after the last kernel (last
            # batch's host work, pruning, final columns)
            "host_exposed_ms_per_step": round(elapsed / steps * 1e3 - main_ms, 2) if world == 1 else None,
            # rank 0: from the moment its last batch's kernels were seen done to the end of the step (the last batch's
            # host work, the pruning with its exchanges, the final columns); per rank in `ranks`
            "tail_after_last_kernel_ms": round(step_timers["tail_ms"] / steps, 2),
            "host_step_parts_ms": {"detect_blobs_sub_rois": round(step_timers["detect_ms"] / steps, 3),
                                   "prune_blobs_mp_and_final_columns": round(step_timers["prune_ms"] / steps, 3)},
            # of which: the LoG stream waiting for a batch's preprocessing on the other stream (HIP events around the waits)
            "pre_stream_wait_ms": None if pre_wait_ms is None else round(pre_wait_ms, 2),
            "overlapped_streams": list(overlapped) or None},
        "above_contract_roofline": flags or None,
        "above_contract_roofline_note": None if not flags else (
            "fractions above 1 are quoted on SURVEY.md 8d's byte count (50 B per voxel and sigma: three unfused passes "
            "with float32 intermediates); the fused kernels with 16-bit intermediates move about 14 B: the contract "
            "roofline is saturated and no longer discriminates, see roofline.actual_frac / valu_busy / mfma_busy"),
        "h2d": None if h2d_gbps is None else {
            "GBps_measured": round(h2d_gbps, 1), "volume_GB": round(vol_bytes / 1e9, 2),
            "ms_for_volume": round(vol_bytes / (h2d_gbps * 1e9) * 1e3, 1),
            "note": "pinned host -> device copy rate measured in this run (256 MiB x 4), extrapolated to the volume; "
                    "the timed region starts with the volume resident (contract), a caller handing a host "
                    "volume pays this once per volume"},
        "kernels": per_kernel,
        "ranks": per_rank,
        # --share k/N: a MODEL of rank k's step in an N-GPU run, measured on one GPU -- its blocks, its seam rows, the
        # pruning of its rows between the recorded seam rows of its neighbours, the merge of every rank's recorded
        # survivors: everything but the transfer (`value` = the whole volume / this step: what N such ranks would give)
        "share": None if not share else {
            "rank": eff_rank, "of": eff_world, "blocks": hi - lo, "batches": list(bl.LAST_BATCH_SIZES),
            "step_ms": round(elapsed / steps * 1e3, 3),
            "kernels_ms": round(gpu_ms, 3), "main_stream_kernels_ms": round(main_ms, 3),
            "start_ms": round(step_timers["start_ms"] / steps, 3),
            "last_batch_host_ms": round(step_timers["last_batch_host_ms"] / steps, 3),
            "prune_and_merge_ms": round(step_timers["prune_ms"] / steps, 3),
            "loopback_copy_ms": round(step_timers["gather_ms"] / steps, 3),
            "tail_after_last_kernel_ms": round(step_timers["tail_ms"] / steps, 3),
            "linear_ms": None if not ctx.get("full_step_ms") else round(ctx["full_step_ms"] / eff_world, 3),
            "note": "MODEL, one GPU: no RCCL transfer in it; the merged table is the whole stack's (table_sha1)"},
        "detector_stats": {k: (round(float(v), 9) if isinstance(v, (float, np.floating)) else int(v))
                           for k, v in vars(stats).items()},
        # the analytic bound of the 16-bit intermediates' rounding error (mmx_tiled_q16_error_bound x value range) that
        # detector_stats.max_f32_error -- the largest |float32 - float64| over all re-scored candidates -- must stay
        # under, and the nomination band that covers it fourfold
        "q16_bound": None if zx_path != nat.MMX_ZX_TILED_Q16 else round(bl.LAST_Q16_BOUND, 9),
        # (the same number under the name the contract uses: error bound of the 16-bit intermediates in VALUE units;
        #  16-bit tiles are chosen only while it is <= log_abs_tol, MMX_LOG_ABS_TOL)
        "q16_bound_abs": None if zx_path != nat.MMX_ZX_TILED_Q16 else round(bl.LAST_Q16_BOUND, 9),
        "log_abs_tol": bl.LOG_ABS_TOL,
        "nms_band": round(bl.LAST_NMS_BAND if zx_path == nat.MMX_ZX_TILED_Q16 else bl.EPS_REL, 9),
        "cpu_baseline": cpu, "parity_sample_identical": parity,
        "parity_sample_source": None if baseline is None else (
            "committed oracle table " + baseline["committed"] if baseline.get("committed") else
            "the oracle, run in this process before the GPU was initialised"),
        "colocs_sha1": None if colocs is None else hashlib.sha1(np.ascontiguousarray(colocs).tobytes()).hexdigest(),
        "scipy": ctx.get("scipy"),
        "volume_gen_s": round(t_gen, 2),
    }
    if host_run is not None:
        out["from_host"] = host_run(out)
        bl.release_buffers()
        torch.cuda.empty_cache()
    if args.dump and name == args.config:
        np.savez(args.dump, final=np.zeros((0, 8)) if final is None else final,
                 colocs=np.zeros((0, n_chl), dtype=np.uint8) if colocs is None else colocs)
    return out


def run_gpu_tiles(args, baseline, steps, warmup, ctx):
    """configs[4] as what it is -- a TILED stack -- with the tiles sharded over the ranks (``--config c5 --tiles T
    [--gpus N]``): every rank holds T tiles resident in HBM (tile g of the stack comes from seed + 2 g: tile 0 is the
    volume the one-GPU c5 record detects) and one step detects each rank's tiles through
    ``stack_detect.detect_blobs_tiles(shard="tiles")`` -- one whole-image detection per tile, as the reference makes one
    call per file (stack_detect.py:338-517), no collective inside the timed region.  Per-GPU work is fixed as N grows:
    ``"scaling": "weak"``; ``value`` = the voxels all ranks processed / the slowest rank's time."""
    import torch
    import torch.distributed as tdist
    from magellanmapper_amd import _native as nat
    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import config, dist, stack_detect, synth
    rank, world, dev, backend = ctx["rank"], ctx["world"], ctx["dev"], ctx["backend"]
    name = args.config
    cfg, PROFILE, shape, n_chl, coloc = config_setup(name, args, None)
    channels = list(range(n_chl))
    config.resolutions = RESOLUTIONS
    config.filename = "bench"
    config.setup_roi_profiles(None)
    config.roi_profile.update(PROFILE)
    for p in config.roi_profiles:
        p.update(PROFILE)
    config.near_max = [-1.0] * max(1, n_chl)
    per_rank, total = int(args.tiles), int(args.tiles) * world
    mine = dist.tile_share(total, rank, world)
    t_gen = time.time()
    resident = {}
    for g in mine:
        vol = synth.make_volume_device(shape, cfg["seed"] + 2 * g, dev)
        if n_chl > 1:
            c1 = synth.make_volume_device(shape, cfg["seed"] + 2 * g + 1, dev)
            c1 = torch.maximum(c1.to(torch.int32), (vol.to(torch.int32) * 7) // 10).to(vol.dtype)
            vol = torch.stack((vol, c1), dim=-1).contiguous()
            del c1
        resident[g] = bl.DeviceVolume(vol, dev)
        del vol
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    # every tile of the stack as the image object the reference's callers hand over; a rank's own tiles are already on
    # its device (`device_volume`), the others are never touched.  (`img`: the (t, z, y, x[, c]) geometry only -- a
    # zero-stride view, no host copy of a resident benchmark tile exists)
    geometry = np.broadcast_to(np.zeros(1, dtype=np.uint16), (1,) + tuple(shape) + ((n_chl,) if n_chl > 1 else ()))
    tiles = [stack_detect.Image5d(geometry) for _ in range(total)]
    if args.budget_gb > 0:
        budget = int(args.budget_gb * (1 << 30))
    else:
        free_b, _ = torch.cuda.mem_get_info()
        sharers = max(1, -(-world // max(1, torch.cuda.device_count()))) if backend != "nccl" else 1
        budget = min(16 << 30, int(0.55 * free_b / sharers))
    bl.blob_log_blocks = functools.partial(ctx["blob_log_blocks"], budget_bytes=budget)
    bl.BUDGET_BYTES = budget               # (calls that do not name a budget: the multi-channel pipeline)

    parity = None
    cpu = None
    if baseline is not None and rank == 0:
        cpu, cpu_final, sample = baseline["cpu"], baseline["final"], baseline["sample"]
        with dist.solo():
            _, _, got = stack_detect.detect_blobs_blocks("bench", stack_detect.Image5d(sample[None]), None, None,
                                                         channels, False, False, True, coloc)
        parity = bool(cpu_final is not None and got.blobs is not None and got.blobs.shape == cpu_final.shape and
                      np.array_equal(canon(got.blobs), canon(cpu_final)))
        if parity and baseline.get("colocs") is not None:
            parity = bool(np.array_equal(got.blobs, cpu_final) and got.colocalizations is not None and
                          np.array_equal(got.colocalizations, baseline["colocs"]))

    tile_ms = {g: [] for g in mine}

    def one_step():
        for g in mine:
            tiles[g].device_volume = resident[g]
        out = []
        t_prev = time.perf_counter()
        for g, blobs in stack_detect.detect_blobs_tiles("bench", tiles, channels, coloc, shard="tiles"):
            now = time.perf_counter()
            tile_ms[g].append((now - t_prev) * 1e3)
            t_prev = now
            out.append((g, blobs))
        return out

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    for g in mine:
        tile_ms[g].clear()
    nat.timing_read()
    nat.timing_enable(args.kernel_events != "none")
    barrier()
    t0 = time.perf_counter()
    results = []
    for _ in range(steps):
        results = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = nat.timing_read()
    nat.timing_enable(False)
    if args.kernel_events == "none":
        nat.timing_enable(True)
        one_step()
        torch.cuda.synchronize()
        ktimes = {k: (ms * steps, n * steps) for k, (ms, n) in nat.timing_read().items()}
        nat.timing_enable(False)
    # after the timed region: every rank's tables on every rank (two small all-gathers), timed apart
    t_g = time.perf_counter()
    everything = stack_detect.gather_tiles(results)
    gather_ms = (time.perf_counter() - t_g) * 1e3

    def sha(a):
        return None if a is None else hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    digests = [(g, sha(b.blobs), sha(b.colocalizations), 0 if b.blobs is None else int(len(b.blobs))) for g, b in everything]
    per_rank_rec = None
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        own_ms = float(np.mean([np.mean(v) for v in tile_ms.values()])) if tile_ms else 0.0
        mine_t = torch.tensor([elapsed / steps * 1e3, own_ms, sum(ms for ms, n in ktimes.values()) / steps, float(len(mine)),
                               gather_ms], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine_t) for _ in range(world)]
        tdist.all_gather(allr, mine_t)
        per_rank_rec = [dict(zip(("step_ms", "ms_per_tile", "kernel_ms", "tiles", "gather_ms"),
                                 (round(float(v), 2) for v in r.cpu()))) for r in allr]
        elapsed = float(t.item())
    resident.clear()
    bl.release_buffers()
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    nvox = int(np.prod(shape))
    ns = PROFILE["num_sigma"]
    my_vox = nvox * n_chl * len(mine)
    per_kernel = {}
    for k, (ms, n) in ktimes.items():
        if n:
            per_kernel[k] = {"ms_per_step": round(ms / steps, 3), "launches_per_step": n // steps}
            if k in ALG_BYTES:
                per_kernel[k]["alg_GBps"] = round(ALG_BYTES[k] * my_vox * ns * steps / (ms * 1e-3) / 1e9, 1)
    stream_k = {k: v for k, v in per_kernel.items() if k in ALG_BYTES}
    dom = max(stream_k, key=lambda k: stream_k[k]["ms_per_step"]) if stream_k else None
    roof = None
    if dom:
        launches = ktimes[dom][1]
        roof = {"bound": "hbm", "kernel": dom, "achieved": stream_k[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(stream_k[dom]["alg_GBps"] / HBM_PEAK_GBS, 4),
                "alg_bytes_per_launch": int(ALG_BYTES[dom] * my_vox * ns * steps / max(1, launches)),
                "avg_launch_ms": round(ktimes[dom][0] / max(1, launches), 4), "traffic": None,
                "copy_GBps": ctx.get("copy_gbps"),
                "note": "rank 0's launches of the dominant LoG kernel over the timed region (HIP events on its launch stream); "
                        "achieved = SURVEY.md 8d's algorithmic bytes of this pass / that time; no counter pass was taken for "
                        "this command (traffic null): the kernel is the one profiles/r06_pmc_counters.json describes, here on "
                        "preprocessed float voxels"}
    ms_tile = elapsed / steps * 1e3 / max(1, per_rank)
    b_alg = B_ALG_PER_SIGMA * ns * n_chl
    tile0 = next((d for d in digests if d[0] == 0), None)
    return {
        "metric": cfg["metric"] + f"; tiled stack, {per_rank} tile(s) per GPU sharded by tile",
        "value": round(nvox * total * steps / elapsed / 1e6, 2), "unit": "Mvoxels/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 preprocessing; " + ZX_DTYPES.get(bl.LAST_ZX_PATH, "f32; f64 re-score of every candidate")
                 if PROFILE["denoise_size"] else ZX_DTYPES.get(bl.LAST_ZX_PATH, "f32; f64 re-score of every candidate"),
        "data": "synthetic",
        "config": {"workload": f"{name} tiled: {total} tiles of {shape[2]}x{shape[1]}x{shape[0]} (x,y,z) x {n_chl} channel(s) "
                               f"uint16 (tile g: seeds {cfg['seed']} + 2 g [, + 1]), {cfg['what']}; one whole-image detection "
                               "per tile (detect + prune + final table), tiles resident in HBM",
                   "tiles_per_rank": per_rank, "tiles_total": total,
                   "parallelism": f"tiles r, r + N, ... on rank r of {world}; no collective in the timed region",
                   "batch_budget_GB": round(budget / (1 << 30), 1)},
        "ms_per_tile": round(ms_tile, 2),
        "Mvoxels_per_s_per_gpu": round(nvox * per_rank * steps / elapsed / 1e6, 2),
        "ranks": per_rank_rec,
        "tile_ms_rank0": {str(g): round(float(np.mean(v)), 2) for g, v in tile_ms.items() if v},
        "tiles": [{"tile": g, "table_sha1": a, "colocs_sha1": c, "blobs": n} for g, a, c, n in digests],
        "tile0_is_the_c5_record_volume": None if tile0 is None else {"table_sha1": tile0[1], "colocs_sha1": tile0[2],
                                                                     "blobs": tile0[3]},
        "blobs": int(sum(d[3] for d in digests)),
        "blobs_per_s": round(sum(d[3] for d in digests) * steps / elapsed, 1),
        "gather_tiles_ms_rank0": round(gather_ms, 2),
        "roofline": roof,
        "pipeline_roofline": {"alg_bytes_per_voxel": b_alg,
                              "gpu_kernel_ms_per_step_rank0": round(sum(ms for ms, n in ktimes.values()) / steps, 2),
                              "frac_wall": round(b_alg * nvox * per_rank / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels": per_kernel,
        "cpu_baseline": cpu, "parity_sample_identical": parity,
        "parity_sample_source": None if baseline is None else (
            "committed oracle table " + baseline["committed"] if baseline.get("committed") else
            "the oracle, run in this process before the GPU was initialised"),
        "scipy": ctx.get("scipy"), "volume_gen_s": round(t_gen, 2),
    }


def from_host_record(kind, slab, z0, shape, volume_cls, detect_and_prune, blocks, rec, n_steps, n_tiles=1):
    """End to end from a HOST volume, what a drop-in caller pays per stack (the reference's callers hand a memory-mapped
    image5d.npy, importer.py:794): every step uploads the volume again -- z-slab by z-slab on a copy stream, the
    detection starting on the blocks whose slabs have landed (blob_log._SlabUpload) -- detects and prunes.
    `kind`: pinned (a pinned tensor: the DMA reads it directly), pageable (an ordinary array: staged through pinned
    buffers by a few host threads) or mmap (a memory-mapped .npy under /dev/shm)."""
    import tempfile
    import torch
    path = None
    pinned = slab                 # (a pinned host copy of the volume, made before the device one was released)
    if kind == "pinned":
        src = pinned
    elif kind == "pageable":
        src = np.array(pinned.numpy())
        del pinned
    else:
        fd, path = tempfile.mkstemp(suffix=".npy", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        os.close(fd)
        np.save(path, pinned.numpy())
        del pinned
        src = np.load(path, mmap_mode="r")
    times, digest, n = [], None, 0
    tiled = None
    try:
        if n_tiles > 1:
            # consecutive tiles of a tiled stack (configs[4]): tile k + 1 is queued for upload before tile k is detected
            # (stack_detect.detect_blobs_tiles does the same with Image5d files); every tile here is the same host volume
            detect_and_prune(volume_cls(src, z0, shape), blocks)          # (staging buffers, retained blocks: warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hv, same = volume_cls(src, z0, shape), True
            each_tile, t_prev = [], t0
            for k in range(n_tiles):
                nxt = volume_cls(src, z0, shape) if k + 1 < n_tiles else None
                final_k, _, _ = detect_and_prune(hv, blocks)
                each_tile.append(round((time.perf_counter() - t_prev) * 1e3, 1))
                t_prev = time.perf_counter()
                d_k = None if final_k is None else hashlib.sha1(np.ascontiguousarray(final_k).tobytes()).hexdigest()
                same = same and d_k == rec.get("table_sha1")
                hv = nxt
            torch.cuda.synchronize()
            ms_tile = (time.perf_counter() - t0) * 1e3 / n_tiles
            h2d_t = (rec.get("h2d") or {}).get("ms_for_volume")
            tiled = {"tiles": n_tiles, "ms_per_tile": round(ms_tile, 2), "ms_each_tile": each_tile,
                     "every_tile_same_table_as_resident": same,
                     "serial_ms_per_tile": None if h2d_t is None else round(h2d_t + rec["ms_per_step"], 2),
                     "ratio_to_max_of_both": None if h2d_t is None else round(ms_tile / max(h2d_t, rec["ms_per_step"]), 3)}
        for i in range(n_steps + 1):               # (the first step also pins the staging buffers: not counted)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hv = volume_cls(src, z0, shape)
            final, _, _ = detect_and_prune(hv, blocks)
            torch.cuda.synchronize()
            if i:
                times.append((time.perf_counter() - t0) * 1e3)
            digest = None if final is None else hashlib.sha1(np.ascontiguousarray(final).tobytes()).hexdigest()
            n = 0 if final is None else len(final)
            del hv
    finally:
        del src
        if path is not None:
            os.unlink(path)
    ms = float(np.median(times))
    h2d = (rec.get("h2d") or {}).get("ms_for_volume")
    floor = max(h2d or 0.0, rec["ms_per_step"])
    nvox_all = int(np.prod(shape))
    return {"source": kind, "ms_per_step": round(ms, 2), "steps": len(times),
            "Mvoxels_per_s": round(nvox_all / (ms * 1e-3) / 1e6, 1),
            # what the host -> device link alone allows for a volume that starts on the host (measured copy rate of this
            # run; uint16 voxels, every channel): no overlap scheme can beat it
            "pcie_ceiling_Mvoxels_per_s": None if h2d is None else round(nvox_all / (h2d * 1e-3) / 1e6, 1),
            "fraction_of_pcie_ceiling": None if h2d is None else round(h2d / ms, 3),
            "resident_ms_per_step": rec["ms_per_step"], "h2d_ms_for_volume": h2d,
            "ratio_to_max_of_both": round(ms / floor, 3) if floor else None,
            "serial_ms": None if h2d is None else round(h2d + rec["ms_per_step"], 2),
            "blobs": n, "table_sha1": digest, "same_table_as_resident": digest == rec.get("table_sha1"),
            "tiled": tiled,
            "note": "every step: upload of the host volume (block row by block row on a copy stream, one event per piece; "
                    "--upload-order slabs: whole z-slabs) + detection of the blocks whose pieces have landed + pruning; "
                    "`serial_ms` = whole upload, then the resident step"}


def compact(rec):
    """The sub-record form of a full record."""
    pr, roof = rec["pipeline_roofline"], rec["roofline"] or {}
    return {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "steps": rec["steps"],
            "ms_per_step": rec["ms_per_step"], "step_ms_percentiles": rec.get("step_ms_percentiles"),
            "blobs": rec["blobs"], "table_sha1": rec["table_sha1"],
            "frac_wall": pr["frac_wall"], "main_stream_kernel_ms_per_step": pr["main_stream_kernel_ms_per_step_rank0"],
            "host_exposed_ms_per_step": pr["host_exposed_ms_per_step"],
            "dominant_kernel": roof.get("kernel"), "dominant_kernel_frac": roof.get("frac"),
            "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in rec["kernels"].items()},
            "parity_sample_identical": rec["parity_sample_identical"],
            "cpu_baseline": None if rec["cpu_baseline"] is None else
            {k: rec["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "pool_note") if k in rec["cpu_baseline"]},
            "config": rec["config"]["workload"], "dtype": rec["dtype"], "graph_replay": rec.get("graph_replay")}


# ------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="workload (default c3; the default command on one GPU also appends compact c2 and c5 sub-records)")
    ap.add_argument("--no-sub-records", action="store_true", help="default command: c3 only")
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="z y x (default: the named config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-events", choices=("all", "dominant", "none"), default="all",
                    help="which kernel families record HIP events inside the timed region (default: all; dominant / none: "
                         "the roofline's kernel only / none, the others timed in extra steps after it)")
    ap.add_argument("--budget-gb", type=float, default=0.0,
                    help="workspace budget per batch (default: 16 GiB = 22 blocks of the benchmark geometry, less when "
                         "the free HBM of this rank's GPU does not allow it; larger batches are SLOWER: the host starts "
                         "on a batch only when its kernels are done and its work is hidden behind the kernels of the "
                         "next ones -- measured 157 / 161 / 165 / 170 ms per volume at 16 / 24 / 32 / 48 GiB)")
    ap.add_argument("--denoise", type=int, default=0, metavar="SIZE",
                    help="per-block preprocessing on (profile denoise_size) for c2 / c3; c5 has it at 25")
    ap.add_argument("--volume", default=None, metavar="NPY",
                    help="a (z, y, x[, c]) uint16 host volume to detect instead of the generated one (parity tests)")
    ap.add_argument("--dump", default=None, metavar="NPZ", help="rank 0 writes the final table (and colocs) here")
    ap.add_argument("--segment-size", type=int, default=0, help="profile segment_size (default 256; parity tests use smaller blocks)")
    ap.add_argument("--cpu-full", action="store_true",
                    help="CPU baseline (and the parity check) on the WHOLE volume of --config instead of the bounded sample")
    ap.add_argument("--cpu-cores", type=int, default=0, help="pool size of the CPU baseline (default: all physical cores)")
    ap.add_argument("--from-host", choices=("pinned", "pageable", "mmap"), default=None,
                    help="after the timed region: steps that start from a HOST copy of the volume (upload overlapped "
                         "with the detection), reported as `from_host`")
    ap.add_argument("--stage-threads", type=int, default=0, help="--from-host pageable / mmap: host threads that fill the pinned staging buffers (0: volume._stage_threads)")
    ap.add_argument("--native-staging", choices=("0", "1"), default="1", help="--from-host pageable / mmap: the staging loop as one native call (1) or the Python loop (0)")
    ap.add_argument("--upload-order", choices=("cells", "slabs"), default="cells",
                    help="--from-host: the volume goes up block row by block row (cells, default) or in z-slabs")
    ap.add_argument("--tiles", type=int, default=1,
                    help="with --from-host: also run this many consecutive tiles, each uploading beside its predecessor's "
                         "detection.  Without --from-host (or with --shard tiles): the workload as a TILED stack, this many "
                         "resident tiles PER GPU, sharded by tile over the ranks -- a weak-scaling line, no collective in "
                         "the timed region (configs[4] across N GPUs)")
    ap.add_argument("--share", default=None, metavar="K/N",
                    help="ONE process stands in for rank K of an N-GPU run of the workload: its share of the blocks, its "
                         "seam rows, the pruning of its rows, the merge of all ranks' survivors, with a recording in place "
                         "of the wire (dist.Loopback) -- a model of the strong-scaling step measured on one GPU")
    ap.add_argument("--full-step-ms", type=float, default=0.0, help="with --share: the one-GPU step to quote linear scaling against")
    ap.add_argument("--pre-stream", choices=("0", "1"), default="1",
                    help="per-block preprocessing on a stream of its own beside the LoG kernels (1, the default) or on the "
                         "LoG stream (0: the kernel families then run one after the other and their HIP-event times are "
                         "each family's ALONE -- what tools/logfloat_profile.sh compares with the raw-voxel run)")
    ap.add_argument("--pre-ahead-retained", type=int, default=-1, help="experiments: blob_log.PRE_AHEAD_RETAINED")
    ap.add_argument("--batches", default=None, metavar="N,N,...", help="experiments: batches of exactly these sizes (blob_log.FORCED_BATCH_SIZES)")
    ap.add_argument("--taper", type=int, default=-1, help="blob_log.TAPER: size of the last batch of a step (experiments)")
    ap.add_argument("--batch-major", choices=("auto", "0", "1"), default="auto",
                    help="several channels: one pipeline, both channels of a batch of blocks before the next batch (1), "
                         "channel after channel (0), or batch-major only while the volume is still uploading (auto, default)")
    ap.add_argument("--stack-finisher", choices=("0", "1"), default="1",
                    help="small one-batch stacks: the host chain behind the kernels as one native call (1, default) or the "
                         "call-by-call form (0)")
    ap.add_argument("--prune-prof", action="store_true", help="print the phases of every pruning step to stderr")
    ap.add_argument("--prune-ahead", choices=("auto", "0", "1"), default="auto",
                    help="prune finished regions while the GPU detects: auto = stacks of 64 blocks and more (stack_detect.PRUNE_AHEAD)")
    ap.add_argument("--shard", choices=("blocks", "tiles"), default=None,
                    help="N > 1: blocks of ONE volume over the ranks (strong scaling, the default) or whole tiles of a "
                         "tiled stack (weak scaling; implied by --tiles T without --from-host)")
    ap.add_argument("--region-split", type=int, default=0,
                    help="regions of the pruning ahead per x-row of blocks (stack_detect.REGION_SPLIT; 0 = one per channel)")
    ap.add_argument("--parity-sample", default=None, metavar="NPZ",
                    help="check the parity sample against this committed oracle table (tests/golden/make_bench_samples.py) "
                         "instead of running the oracle: no cpu_baseline timing, seconds instead of minutes")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # N ranks wanted, none launched: launch them (before anything here touches the GPU)
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    explicit = args.config is not None
    args.config = args.config or "c3"
    by_tile = args.shard == "tiles" or (args.shard is None and args.tiles > 1 and not args.from_host)
    if by_tile and (args.volume or args.from_host or args.dump):
        raise SystemExit("--shard tiles generates its resident tiles itself: not with --volume / --from-host / --dump")
    host_vol = np.load(args.volume, mmap_mode="r") if args.volume else None

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # before anything touches the GPU runtime (the host driver only supports dmabuf IPC)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    subs = [] if (explicit or args.no_sub_records or world > 1 or args.volume or args.shape or args.denoise
                  or args.segment_size or args.dump or by_tile or args.share) else list(SUB_RECORDS)

    import torch
    import torch.distributed as tdist

    # ---------------- CPU baselines (rank 0, N = 1 only) BEFORE the GPU is initialised: the worker
    # pool is spawned (fork + exec), which must not happen from a process that holds a GPU context
    # (N > 1: only where no committed oracle table fits the workload -- the other ranks then wait for rank 0 in
    #  init_process_group, whose timeout allows for that)
    baselines = {}
    # host allocator, before anything large is allocated: the per-step tables (tens of MB) come from the heap and stay
    # mapped between steps instead of being mmap'd, page-faulted in and unmapped every step (8 ms of a 198 ms step,
    # tools/steptrace.py)
    from magellanmapper_amd import _native as nat
    nat.keep_host_heap()
    if rank == 0 and world > 1 and not args.parity_sample and not args.no_cpu_baseline and not args.cpu_full and not (
            args.volume or args.shape or args.denoise or args.segment_size):
        # the CPU baseline is timed at N = 1 only (the contract's wording): the lines of N > 1 check parity against the
        # COMMITTED oracle table of the same sample instead (no minute of oracle work with N - 1 ranks waiting)
        committed = os.path.join(ROOT, "tests", "golden", f"bench_sample_{args.config}.npz")
        if os.path.exists(committed):
            args.parity_sample = committed
    if rank == 0 and args.parity_sample:
        got = load_parity_sample(args.parity_sample, args.config, args)
        if got is not None:
            baselines[args.config] = got
    if rank == 0 and not args.no_cpu_baseline:
        for name in [args.config] + subs:
            if name not in baselines:
                baselines[name] = run_cpu_baseline(name, args, host_vol)
    if any(b.get("cpu") for b in baselines.values()):
        # the oracle pools have just ended: a hundred-odd spawned interpreters are still being torn down and the cores
        # they loaded for a minute are hot.  Two runs of this command on two boxes read 104.3 / 104.9 ms with 10.8 / 11.9
        # ms of exposed host time when the GPU part followed at once (100.0 without the pools on the same box); with this
        # pause (and the allocator set up first) 99.1 / 99.3 against 99.8 / 101.0 alternating (round 4)
        time.sleep(float(os.environ.get("MMX_BENCH_SETTLE_S", 3)))
    # one rank per GPU; MMX_DIST_BACKEND=gloo + fewer GPUs than ranks is only for functional tests
    backend = os.environ.get("MMX_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: --gpus {world} needs one GPU per rank, this node shows {n_dev} "
                         "(a functional run of the N-rank path on fewer GPUs: MMX_DIST_BACKEND=gloo)")
    local_dev = local_rank % max(1, n_dev) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        from datetime import timedelta
        wait = timedelta(minutes=30)        # rank 0 arrives after its CPU baseline (about a minute on 128 cores)
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=dev, timeout=wait)
        else:
            tdist.init_process_group(backend, timeout=wait)

    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import stack_detect as _sd
    from magellanmapper_amd import stack_prune as _sp
    _sp.PRUNE_PROF = bool(args.prune_prof)
    _sd.STACK_FINISHER = args.stack_finisher == "1"
    if args.batch_major != "auto":
        bl.BATCH_MAJOR = args.batch_major == "1"
    if args.taper >= 0:
        bl.TAPER = args.taper
    if args.pre_ahead_retained >= 0:
        bl.PRE_AHEAD_RETAINED = args.pre_ahead_retained
    if args.batches:
        bl.FORCED_BATCH_SIZES = [int(v) for v in args.batches.split(",")]
    bl.PRE_STREAM = args.pre_stream == "1"
    from magellanmapper_amd import volume as _volume
    if args.stage_threads:
        _volume._STAGE_THREADS = args.stage_threads
    _volume.NATIVE_STAGING = args.native_staging == "1"
    if args.prune_ahead != "auto":
        _sd.PRUNE_AHEAD = args.prune_ahead
    if args.region_split:
        _sd.REGION_SPLIT = args.region_split

    import scipy
    share = wire = None
    if args.share:
        if world > 1 or by_tile:
            raise SystemExit("--share models ONE rank in ONE process: not with --gpus N > 1 or --shard tiles")
        from magellanmapper_amd import dist as _dist
        k_, n_ = (int(v) for v in args.share.split("/"))
        share, wire = (k_, n_), _dist.Loopback(k_, n_)
        _dist.set_loopback(wire)
    ctx = dict(rank=rank, world=world, dev=dev, backend=backend, blob_log_blocks=bl.blob_log_blocks,
               share=share, wire=wire, full_step_ms=args.full_step_ms,
               # the one third-party routine whose implementation-defined order reaches the result (chain blocks of the
               # overlap prune: blob_log._reference_pair_order); fixtures were made with 1.7.1, tests run with this one
               scipy=scipy.__version__)

    # ---------------- measured beside the roofline peak: a device copy and the host -> device link
    if rank == 0:
        n = 1 << 28                                   # 1 GiB in + 1 GiB out: far beyond the 256 MiB Infinity Cache
        a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        s_ptr = torch.cuda.current_stream().cuda_stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for it in range(3):
            if it == 1:
                ev[0].record()
            nat.check(nat.lib().mmx_calib_stream(1, a.data_ptr(), b.data_ptr(), n, s_ptr), "mmx_calib_stream")
        ev[1].record()
        torch.cuda.synchronize()
        ctx["copy_gbps"] = round(2 * 2 * n * 4 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, 1)
        hbuf = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()       # 256 MiB pinned
        dbuf = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
        dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(4):
            dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()
        ctx["h2d_gbps"] = 4 * (1 << 28) / (time.perf_counter() - t_h) / 1e9
        del a, b, hbuf, dbuf
        torch.cuda.empty_cache()

    if by_tile:
        out = run_gpu_tiles(args, baselines.get(args.config), args.steps, args.warmup, ctx)
    else:
        out = run_gpu(args.config, args, host_vol, baselines.get(args.config), args.steps, args.warmup, ctx)
    if subs and out is not None:
        out["sub_records"] = {}
        for name in subs:
            st, wu = SUB_RECORDS[name]
            out["sub_records"][name] = compact(run_gpu(name, args, host_vol, baselines.get(name), st, wu, ctx))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

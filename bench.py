#!/usr/bin/env python3
"""Benchmark of the whole-volume nuclei-detection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs; the default, c3, is the one the metric is quoted on):
  c3  2048 x 2048 x 1024 uint16, 5-sigma LoG scale space + NMS + prune   (configs[2]; N > 1: configs[3])
  c2  512 x 512 x 256 uint16, single-sigma LoG                            (configs[1])
  c5  2-channel 2048 x 2048 x 512 uint16 tile of a tiled light-sheet stack: per-block preprocessing
      (denoise_size 25), detection of both channels, intensity co-localisation, prune (configs[4])
At N > 1 the SAME volume's blocks are sharded over the ranks (strong scaling).  A *step* is one full pass of
the hot path over the volume: all blocks through the HIP kernels, the blob-table gather, the overlap pruning;
the volume is resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel: algorithmic bytes / HIP-event time on its launch stream; the device-copy rate
                measured in this run beside the 8 TB/s peak
  cpu_baseline  the oracle (NumPy/SciPy restatement of the reference) on this box's host cores over a bounded
                sample of the same workload, detection and pruning seconds apart
  ranks         per-rank kernel / gather / prune milliseconds (N > 1)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import multiprocessing as mp
import os
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
RESOLUTIONS = np.array([[1.0, 1.0, 1.0]])
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
#: algorithmic HBM bytes per voxel per sigma of each kernel
ALG_BYTES = {"zpass": 2 + 8, "ypass": 8 + 8, "xpass": 8 + 4, "peaks": 4,
             # fused path (default): Z+X in one kernel (Gz / Gzz never leave the CU), then Y
             "zxpass": 2 + 8, "y2pass": 8 + 4}
B_ALG_PER_SIGMA = 50                # SURVEY.md section 8d contract figure


# ------------------------------------------------------------------ CPU baseline (oracle)
def _cpu_block(args):
    """One block through the oracle (runs in a spawned worker: NumPy/SciPy only)."""
    coord, offset, last_coord, sub, profiles, dms, channel, coloc = args
    near_max = [-1.0] * max(1, len(profiles))
    from oracle import magmap_oracle as mmo
    return coord, mmo.detect_sub_roi(coord, offset, last_coord, None, sub, channel, profiles,
                                     RESOLUTIONS, denoise_max_shape=dms, near_max=near_max, coloc=coloc)


def cpu_baseline(sample: np.ndarray, cores: int, profile: dict, channels, coloc: bool):
    """Reference strategy (magmap/cv/stack_detect.py:222-257): a process pool over blocks."""
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
    final = None
    if pruned is not None:
        pruned[:, 0:3] = pruned[:, 7:10]
        final = pruned[:, [0, 1, 2, 3, 4, 5, 6, 10]]
    return final, t_detect, time.time() - t0, len(jobs)


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


# ------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="z y x (default: the named config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    ap.add_argument("--cpu-cores", type=int, default=0, help="pool size of the CPU baseline (default: all physical cores)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    PROFILE = dict(_BASE_PROFILE, **cfg["profile"])
    if args.denoise:
        PROFILE["denoise_size"] = args.denoise
    if args.segment_size:
        PROFILE["segment_size"] = args.segment_size
    host_vol = np.load(args.volume, mmap_mode="r") if args.volume else None
    shape = tuple(host_vol.shape[:3]) if host_vol is not None else (tuple(args.shape) if args.shape else cfg["shape"])
    n_chl = (host_vol.shape[3] if host_vol.ndim > 3 else 1) if host_vol is not None else cfg["channels"]
    channels = list(range(n_chl))
    coloc = bool(cfg["coloc"] and n_chl > 1)
    seed = cfg["seed"]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # before anything touches the GPU runtime (the host driver only supports dmabuf IPC)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as tdist

    def make_host_sample(shp):
        """Host sample of the workload from the same generator (per-channel seeds, channel 1 shares 70 % of channel 0)."""
        from magellanmapper_amd import synth as _synth
        cpu_dev = torch.device("cpu")
        c0 = _synth.make_volume_device(shp, seed, cpu_dev).to(torch.int32)
        if n_chl == 1:
            return c0.numpy().astype(np.uint16)
        c1 = _synth.make_volume_device(shp, seed + 1, cpu_dev).to(torch.int32)
        c1 = torch.maximum(c1, (c0 * 7) // 10)
        return torch.stack((c0, c1), dim=-1).numpy().astype(np.uint16)

    # ---------------- CPU baseline (rank 0, N = 1 only) BEFORE the GPU is initialised: the worker
    # pool is spawned (fork + exec), which must not happen from a process that holds a GPU context
    cpu = None
    cpu_final = None
    sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        phys = physical_cores()
        cores = args.cpu_cores or phys
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
        sample = make_host_sample(sshape) if host_vol is None else np.ascontiguousarray(
            host_vol[:sshape[0], :sshape[1], :sshape[2]])
        cpu_final, t_det, t_tot, n_jobs = cpu_baseline(sample, cores, PROFILE, channels, coloc)
        cpu = {"value": round(int(np.prod(sshape)) / t_tot / 1e6, 3), "unit": "Mvoxels/s",
               "cores": min(cores, n_jobs), "kind": "port",
               "cpu_count": os.cpu_count(), "physical_cores": phys,
               "detection_s": round(t_det, 2), "pruning_s": round(t_tot - t_det, 2),
               "sample": f"{sshape[0]}x{sshape[1]}x{sshape[2]} (z,y,x){' x %d channels' % n_chl if n_chl > 1 else ''} "
                         f"volume from the same generator (seed, blob density, profile, segment_size as the GPU run), "
                         f"{n_jobs} blocks over a pool of {min(cores, n_jobs)} processes "
                         f"(reference strategy, stack_detect.py:222-257)",
               "blobs": 0 if cpu_final is None else int(len(cpu_final))}
    # one rank per GPU; MMX_DIST_BACKEND=gloo + fewer GPUs than ranks is only for functional tests
    backend = os.environ.get("MMX_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    local_dev = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=dev)
        else:
            tdist.init_process_group(backend)

    from magellanmapper_amd import _native as nat
    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import config, detector, dist, stack_detect, synth

    # host allocator: the per-step tables (tens of MB) come from the heap and stay mapped between steps instead
    # of being mmap'd, page-faulted in and unmapped every step (8 ms of a 198 ms step, tools/steptrace.py)
    nat.keep_host_heap()

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
    lo, hi = dist.share_bounds(n_blocks, rank, world)
    zs = [blocks.sub_roi_slices[coords[i]][0] for i in range(lo, hi)]
    z0 = min(s.indices(shape[0])[0] for s in zs) if zs else 0
    z1 = max(s.indices(shape[0])[1] for s in zs) if zs else 1
    t_gen = time.time()
    if host_vol is not None:
        slab = torch.from_numpy(np.ascontiguousarray(host_vol[z0:z1]).view(np.int16)).to(dev).view(torch.uint16)
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
    class SlabVolume(bl.DeviceVolume):
        def __init__(self, t, z_off, full_shape):
            super().__init__(t, dev)
            self.z_off = z_off
            self.shape = tuple(full_shape) + tuple(t.shape[3:])

        def view(self, channel, for_f32_passes):
            v = super().view(channel, for_f32_passes)
            v.d_data = v.d_data - self.z_off * self.tensor.stride()[0] * self.tensor.element_size()
            return v

    dvol = SlabVolume(slab, z0, shape)
    timers = {"gather_ms": 0.0, "prune_ms": 0.0, "detect_ms": 0.0}

    def finish(pruned):
        if pruned is None:
            return None, None
        bb = detector.Blobs(pruned)              # the table's final form (reference stack_detect.py:458-467)
        bb.replace_rel_with_abs_blob_coords(pruned)
        colocs = pruned[:, 10:10 + n_chl].astype(np.uint8) if coloc else None
        return bb.remove_abs_blob_coords(True), colocs

    def one_step():
        t_a = time.perf_counter()
        seg = stack_detect.StackDetector.detect_blobs_sub_rois(
            None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets, blocks.denoise_max_shape,
            blocks.exclude_border, coloc, channels)
        st = stack_detect.StackDetector.last_stats
        t_b = time.perf_counter()
        final = colocs = None
        if rank == 0:
            pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
                dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets,
                channels, blocks.overlap_padding)
            final, colocs = finish(pruned)
        t_c = time.perf_counter()
        timers["gather_ms"] += dist.last_gather_ms()
        timers["detect_ms"] += (t_b - t_a) * 1e3 - dist.last_gather_ms()
        timers["prune_ms"] += (t_c - t_b) * 1e3
        return final, colocs, st

    # workspace budget per batch: from the free HBM of this GPU unless given (ranks that share a GPU in the
    # functional tests share its memory)
    import functools
    if args.budget_gb > 0:
        budget = int(args.budget_gb * (1 << 30))
    else:
        free_b, _ = torch.cuda.mem_get_info()
        sharers = max(1, -(-world // max(1, torch.cuda.device_count()))) if backend != "nccl" else 1
        budget = min(16 << 30, int(0.55 * free_b / sharers))
    bl.blob_log_blocks = functools.partial(bl.blob_log_blocks, budget_bytes=budget)

    # ---------------- the HIP path on the CPU-baseline sample must give the identical table
    parity = None
    if cpu is not None:
        sblocks = stack_detect.setup_blocks(config.roi_profile, sample.shape[:3])
        sdvol = bl.DeviceVolume(sample)
        seg = stack_detect.StackDetector.detect_blobs_sub_rois(
            None, sdvol, sblocks.sub_roi_slices, sblocks.sub_rois_offsets, sblocks.denoise_max_shape,
            None, coloc, channels)
        pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
            sdvol, seg, sblocks.overlap, sblocks.tol, sblocks.sub_roi_slices, sblocks.sub_rois_offsets,
            channels, sblocks.overlap_padding)
        gpu_final, _ = finish(pruned)
        parity = bool(cpu_final is not None and gpu_final is not None and gpu_final.shape == cpu_final.shape and
                      np.array_equal(canon(gpu_final), canon(cpu_final)))
        del sdvol

    # ---------------- measured beside the roofline peak: a device copy and the host -> device link
    copy_gbps = h2d_gbps = None
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
        copy_gbps = 2 * 2 * n * 4 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
        hbuf = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()       # 256 MiB pinned
        dbuf = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
        dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(4):
            dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()
        h2d_gbps = 4 * (1 << 28) / (time.perf_counter() - t_h) / 1e9
        del a, b, hbuf, dbuf

    # ---------------- warm-up, then the timed region
    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    for k in timers:
        timers[k] = 0.0
    nat.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    final = colocs = None
    stats = None
    for _ in range(args.steps):
        final, colocs, stats = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = nat.timing_read()
    nat.timing_enable(False)
    per_rank = None
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([sum(ms for ms, n in ktimes.values()) / args.steps, timers["detect_ms"] / args.steps,
                             timers["gather_ms"] / args.steps, timers["prune_ms"] / args.steps, float(hi - lo)],
                            dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        tdist.all_gather(allr, mine)
        per_rank = [dict(zip(("kernel_ms", "detect_wall_ms", "gather_ms", "prune_ms", "blocks"),
                             (round(float(v), 2) for v in r.cpu()))) for r in allr]

    if rank == 0:
        nvox = int(np.prod(shape))
        ns = PROFILE["num_sigma"]
        my_vox = stats.n_voxels                       # block voxels (all channels) this rank filtered per step
        per_kernel = {}
        for k, (ms, n) in ktimes.items():
            if n:
                per_kernel[k] = {"ms_per_step": round(ms / args.steps, 3), "launches_per_step": n // args.steps}
                if k in ALG_BYTES:
                    gbs = ALG_BYTES[k] * my_vox * ns * args.steps / (ms * 1e-3) / 1e9
                    per_kernel[k]["alg_GBps"] = round(gbs, 1)
                    if k == "peaks" and gbs > HBM_PEAK_GBS:
                        per_kernel[k]["note"] = ("sparse NMS over the entries the Y pass leaves (16 B per 64 voxels and "
                                                 "sigma, plus the lines of the set bits); alg_GBps is quoted on the "
                                                 "4 B contract figure")
                elif k == "preproc":      # once per voxel (not per sigma): 2 B in, 8 + 4 B out; fp64-VALU bound
                    per_kernel[k]["alg_GBps"] = round(14 * my_vox * args.steps / (ms * 1e-3) / 1e9, 1)
        stream_k = {k: v for k, v in per_kernel.items() if k in ALG_BYTES}
        dom = max(stream_k, key=lambda k: stream_k[k]["ms_per_step"]) if stream_k else None
        roof = None
        # HBM bytes per launch of the dominant kernel: NOT measured in this run (PMC passes need rocprofv3); when the
        # committed PMC summary of this same command names the kernel, its figure is quoted with its source
        traffic = traffic_src = None
        try:
            src = "profiles/r02_pmc_traffic.json"
            with open(os.path.join(ROOT, src)) as f:
                pmc = json.load(f)
            if args.config == "c3" and tuple(shape) == cfg["shape"] and world == 1 and dom in pmc["per_launch_GB"]:
                traffic = round(pmc["per_launch_GB"][dom]["total_GB"] * 1e9)
                traffic_src = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, calibrated; not this run)"
        except (OSError, KeyError, ValueError):
            pass
        if dom:
            launches = ktimes[dom][1]
            roof = {"bound": "hbm", "kernel": dom, "achieved": stream_k[dom]["alg_GBps"],
                    "note": ("zxpass = fused Z+X pass on the matrix cores (zx_mode 7: tiled, 16-bit intermediates): the "
                             "10 algorithmic B/voxel/sigma of the contract replace the 22 of the separate Z and X passes "
                             "(+ 0.8 B/voxel/sigma for the operand-ordered voxel copy, 'zxpack', made once per batch); "
                             "the kernel itself moves 6.4 B/voxel because P and Q leave as 16-bit fixed point (error "
                             "bound 4.3e-5, covered fourfold by the NMS band; decisions are taken on exact float64 "
                             "values).  'traffic' below the algorithmic bytes is that.  With float32 tiles it sat at "
                             "the ~80 L2 requests a CU keeps in flight; now its VALU is 58 % and its MFMA pipe 43 % "
                             "busy: instruction issue (DESIGN.md section 4b).  The Y pass of the step runs at "
                             f"{stream_k.get('y2pass', {}).get('alg_GBps', 0) / HBM_PEAK_GBS:.2f} of peak on its "
                             "contract bytes (it reads 16-bit tiles too); see 'kernels' and 'pipeline_roofline'")
                    if dom == "zxpass" else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(stream_k[dom]["alg_GBps"] / HBM_PEAK_GBS, 4),
                    "copy_GBps": None if copy_gbps is None else round(copy_gbps, 1),
                    "frac_of_copy": None if not copy_gbps else round(stream_k[dom]["alg_GBps"] / copy_gbps, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "alg_bytes_per_launch": int(ALG_BYTES[dom] * my_vox * ns * args.steps / max(1, launches)),
                    "avg_launch_ms": round(ktimes[dom][0] / max(1, launches), 4)}
            if dom == "zxpass" and bl.LAST_ZX_PATH in (nat.MMX_ZX_TILED, nat.MMX_ZX_TILED_Q16):
                # the same kernel against the matrix-core roofline: MFMAs it issues (16 x 16 x 32 float16, 16 384 flop
                # each) per 16 x 16 tile step -- X pass 12 (16-bit tiles) or 16 per two k-steps, 6 / 8 for radius <= 8;
                # Z pass 9 per k-step -- over every block row, column tile and z step of this rank's blocks
                from magellanmapper_amd import kernels1d as k1
                space = bl.ScaleSpace.make(PROFILE["min_sigma_factor"] * detector.calc_scaling_factor()[2],
                                           PROFILE["max_sigma_factor"] * detector.calc_scaling_factor()[2], ns)
                q16 = bl.LAST_ZX_PATH == nat.MMX_ZX_TILED_Q16
                flop = 0.0
                for i in range(lo, hi):
                    shp = [s_.indices(n_)[1] - s_.indices(n_)[0] for s_, n_ in zip(blocks.sub_roi_slices[coords[i]], shape)]
                    for R in space.radii:
                        nkx, la = (1, 1) if R <= 8 else ((2, 1) if R <= 16 else (2, 2))
                        per_step = nkx * (6 if q16 else 8) + (la + 1) * 9
                        flop += shp[1] * -(-shp[2] // 16) * (-(-shp[0] // 16) + la) * per_step * 16384.0
                tfs = flop * args.steps / (ktimes[dom][0] * 1e-3) / 1e12
                roof["mfma"] = {"achieved_TFLOPs": round(tfs, 1), "peak_TFLOPs": 2500.0, "frac": round(tfs / 2500.0, 4),
                                "note": "float16 MFMA flops the kernel issues (split-float16 products: 3 MFMAs per float32 "
                                        "product) over its duration, against the dense float16 peak"}
        gpu_ms = sum(ms for ms, n in ktimes.values()) / args.steps
        b_alg = B_ALG_PER_SIGMA * ns * n_chl
        vol_bytes = nvox * n_chl * 2
        out = {
            "metric": cfg["metric"],
            "value": round(nvox * args.steps / elapsed / 1e6, 2), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if host_vol is None else args.volume,
            "config": {"workload": f"{args.config}: {shape[2]}x{shape[1]}x{shape[0]} (x,y,z) uint16 Gaussian-blob volume, "
                                   f"seed {seed}, {n_blocks} blocks (segment_size {PROFILE['segment_size']}, overlap 5), {cfg['what']}, "
                                   "threshold 0.1, overlap 0.5; detect + gather + prune"
                                   + (f"; per-block preprocessing (denoise_size {PROFILE['denoise_size']}: saturate + "
                                      "unsharp + erosion in float64)" if PROFILE["denoise_size"] else ""),
                       "blocks_per_rank": hi - lo, "parallelism": f"blocks sharded over {world} GPU(s)",
                       "batch_budget_GB": round(budget / (1 << 30), 1)},
            "blobs": 0 if final is None else int(len(final)),
            # digest of the final 8-column table of the last step (computed after the timed region): the same
            # for every path / batch size / rank count that is correct (tests compare smaller volumes with the
            # oracle row by row; this is the full-size cross-check)
            "table_sha1": None if final is None else hashlib.sha1(np.ascontiguousarray(final).tobytes()).hexdigest(),
            "blobs_per_s": round((0 if final is None else len(final)) * args.steps / elapsed, 1),
            "roofline": roof,
            "pipeline_roofline": {
                "alg_bytes_per_voxel": b_alg,
                "gpu_kernel_ms_per_step_rank0": round(gpu_ms, 2),
                "achieved_GBps_kernels": round(b_alg * (nvox / world) / (gpu_ms * 1e-3) / 1e9, 1),
                "frac_kernels": round(b_alg * (nvox / world) / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "achieved_GBps_wall": round(b_alg * nvox / (elapsed / args.steps) / 1e9, 1),
                "frac_wall": round(b_alg * nvox / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS / world, 4),
                "host_exposed_ms_per_step": round(elapsed / args.steps * 1e3 - gpu_ms, 2) if world == 1 else None},
            "h2d": None if h2d_gbps is None else {
                "GBps_measured": round(h2d_gbps, 1), "volume_GB": round(vol_bytes / 1e9, 2),
                "ms_for_volume": round(vol_bytes / (h2d_gbps * 1e9) * 1e3, 1),
                "note": "pinned host -> device copy rate measured in this run (256 MiB x 4), extrapolated to the volume; "
                        "the timed region starts with the volume resident (contract), a caller handing a host "
                        "volume pays this once per volume"},
            "kernels": per_kernel,
            "ranks": per_rank,
            "detector_stats": {k: (round(float(v), 9) if isinstance(v, (float, np.floating)) else int(v))
                               for k, v in vars(stats).items()},
            "cpu_baseline": cpu, "parity_sample_identical": parity,
            "volume_gen_s": round(t_gen, 2),
        }
        if args.dump:
            np.savez(args.dump, final=np.zeros((0, 8)) if final is None else final,
                     colocs=np.zeros((0, n_chl), dtype=np.uint8) if colocs is None else colocs)
        print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

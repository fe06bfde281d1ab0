#!/usr/bin/env python3
"""Benchmark of the whole-volume nuclei-detection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mvoxels/s (and blobs/s) on a synthetic 2048 x 2048 x 1024 uint16
stack, 5-sigma LoG scale space + NMS + overlap prune + cross-block de-duplication
(BASELINE.json configs[2] at N = 1, configs[3] at N > 1: the SAME volume, its 4 x 8 x 8
blocks sharded over the ranks -> strong scaling).  A *step* is one full pass of the hot
path over the volume: all blocks through the HIP kernels, the blob-table gather, the
overlap pruning; the volume is resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel: algorithmic bytes / HIP-event time on its launch stream
  cpu_baseline  the oracle (NumPy/SciPy restatement of the reference) on this box's host
                cores over a bounded sample of the same volume
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

SHAPE = (1024, 2048, 2048)          # z, y, x  ("2048 x 2048 x 1024" in x, y, z)
SEED = 3
PROFILE = dict(min_sigma_factor=3, max_sigma_factor=5, num_sigma=5, detection_threshold=0.1,
               overlap=0.5, exclude_border=None, segment_size=256, denoise_size=None,
               prune_tol_factor=(1, 1, 1), isotropic=None,
               # preprocessing keys (only read with --denoise): the reference's defaults
               clip_vmin=5, clip_vmax=99.5, clip_min=0.2, clip_max=1.0, max_thresh_factor=0.5,
               tot_var_denoise=None, unsharp_strength=0.3, erosion_threshold=0.2)
RESOLUTIONS = np.array([[1.0, 1.0, 1.0]])
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s copy-measured)
#: algorithmic HBM bytes per voxel per sigma of each kernel
ALG_BYTES = {"zpass": 2 + 8, "ypass": 8 + 8, "xpass": 8 + 4, "peaks": 4,
             # fused path (default): Z+X in one kernel (Gz / Gzz never leave the CU), then Y
             "zxpass": 2 + 8, "y2pass": 8 + 4}
B_ALG_PER_SIGMA = 50                # SURVEY.md section 8d contract figure


# ------------------------------------------------------------------ CPU baseline (oracle)
def _cpu_block(args):
    """One block through the oracle (runs in a spawned worker: NumPy/SciPy only)."""
    coord, offset, last_coord, sub, profile, dms = args
    from oracle import magmap_oracle as mmo
    return coord, mmo.detect_sub_roi(coord, offset, last_coord, None, sub, None, [profile],
                                     RESOLUTIONS, denoise_max_shape=dms)


def cpu_baseline(sample: np.ndarray, cores: int):
    """Reference strategy (magmap/cv/stack_detect.py:222-257): a process pool over blocks."""
    from oracle import magmap_oracle as mmo
    t0 = time.time()
    blocks = mmo.setup_blocks(PROFILE, sample.shape, RESOLUTIONS)
    sl, off = blocks["sub_roi_slices"], blocks["sub_rois_offsets"]
    last = np.subtract(sl.shape, 1)
    jobs = [(c, off[c], last, sample[sl[c]], PROFILE, blocks["denoise_max_shape"])
            for c in np.ndindex(*sl.shape)]
    seg = np.zeros(sl.shape, dtype=object)
    with mp.get_context("spawn").Pool(processes=cores) as pool:
        for coord, tbl in pool.imap_unordered(_cpu_block, jobs):
            seg[coord] = tbl
    t_detect = time.time() - t0
    pruned, _ = mmo.prune_blobs_mp(sample.shape, seg, blocks["overlap"], blocks["tol"], sl, off, [0],
                                   blocks["overlap_padding"])
    final = None
    if pruned is not None:
        pruned[:, 0:3] = pruned[:, 7:10]
        final = pruned[:, [0, 1, 2, 3, 4, 5, 6, 10]]
    return final, t_detect, time.time() - t0, len(jobs)


def canon(t):
    return t[np.lexsort(tuple(t[:, i] for i in range(t.shape[1] - 1, -1, -1)))]


# ------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="z y x (default: the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--budget-gb", type=float, default=64.0, help="workspace budget per batch")
    ap.add_argument("--denoise", type=int, default=0, metavar="SIZE",
                    help="secondary workload: per-block preprocessing on (profile denoise_size, the "
                         "reference's default is 25); the headline metric is quoted WITHOUT it")
    args = ap.parse_args()
    if args.denoise:
        PROFILE["denoise_size"] = args.denoise
    shape = tuple(args.shape) if args.shape else SHAPE

    import torch
    import torch.distributed as tdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")

    # ---------------- CPU baseline (rank 0, N = 1 only) BEFORE the GPU is initialised: the worker
    # pool is spawned (fork + exec), which must not happen from a process that holds a GPU context
    cpu = None
    cpu_final = None
    sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from magellanmapper_amd import synth as _synth
        cores = max(1, min(os.cpu_count() or 1, 16))
        by = max(1, min(shape[1] // 256, int(np.sqrt(cores))))
        bx = max(1, min(shape[2] // 256, cores // by))
        sz = min(320, shape[0])       # two z-layers of blocks: the sample prunes seams along all three axes
        sample = _synth.make_volume_device((sz, 256 * by, 256 * bx), SEED, torch.device("cpu"))
        sample = sample.to(torch.int32).numpy().astype(np.uint16)
        cpu_final, t_det, t_tot, n_jobs = cpu_baseline(sample, cores)
        cpu = {"value": round(sample.size / t_tot / 1e6, 3), "unit": "Mvoxels/s",
               "cores": min(cores, n_jobs), "kind": "port",
               "sample": f"{sz}x{256 * by}x{256 * bx} (z,y,x) volume from the same generator (seed, blob "
                         f"density, profile, segment_size as the GPU run), {n_jobs} blocks over a pool of "
                         f"{min(cores, n_jobs)} processes; detection {t_det:.1f}s of {t_tot:.1f}s total",
               "blobs": 0 if cpu_final is None else int(len(cpu_final))}
    # one rank per GPU; MMX_DIST_BACKEND=gloo + fewer GPUs than ranks is only for functional tests
    backend = os.environ.get("MMX_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    local_dev = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=dev)
        else:
            tdist.init_process_group(backend)

    from magellanmapper_amd import _native as nat
    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import config, detector, dist, stack_detect, synth

    config.resolutions = RESOLUTIONS
    config.filename = "bench"
    config.setup_roi_profiles(None)
    config.roi_profile.update(PROFILE)
    blocks = stack_detect.setup_blocks(config.roi_profile, shape)
    grid = blocks.sub_roi_slices.shape
    n_blocks = int(np.prod(grid))

    # Each rank needs the z-range its blocks touch (blocks are z-major contiguous per rank).
    coords = list(np.ndindex(*grid))
    lo, hi = dist.share_bounds(n_blocks, rank, world)
    zs = [blocks.sub_roi_slices[coords[i]][0] for i in range(lo, hi)]
    z0 = min(s.start for s in zs) if zs else 0
    z1 = max(s.stop for s in zs) if zs else 1
    t_gen = time.time()
    slab = synth.make_volume_device(shape, SEED, dev, z_range=(z0, z1))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    # A rank-local view that behaves like the full (z, y, x) ROI for block addressing.
    class SlabVolume(bl.DeviceVolume):
        def __init__(self, t, z_off, full_shape):
            super().__init__(t, dev)
            self.z_off = z_off
            self.shape = tuple(full_shape)

        def view(self, channel, for_f32_passes):
            v = super().view(channel, for_f32_passes)
            v.d_data = v.d_data - self.z_off * self.tensor.stride()[0] * self.tensor.element_size()
            return v

    dvol = SlabVolume(slab, z0, shape)

    def one_step():
        seg = stack_detect.StackDetector.detect_blobs_sub_rois(
            None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets, blocks.denoise_max_shape, None,
            False, [0])
        st = stack_detect.StackDetector.last_stats
        final = None
        if rank == 0:
            pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
                dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets,
                [0], blocks.overlap_padding)
            if pruned is not None:      # the table's final form (reference stack_detect.py:458-467)
                bb = detector.Blobs(pruned)
                bb.replace_rel_with_abs_blob_coords(pruned)
                final = bb.remove_abs_blob_coords(True)
        return final, st

    # monkey: budget for the workspace
    import functools
    bl.blob_log_blocks = functools.partial(bl.blob_log_blocks, budget_bytes=int(args.budget_gb * (1 << 30)))

    # ---------------- the HIP path on the CPU-baseline sample must give the identical table
    parity = None
    if cpu is not None:
        sblocks = stack_detect.setup_blocks(config.roi_profile, sample.shape)
        sdvol = bl.DeviceVolume(sample)
        seg = stack_detect.StackDetector.detect_blobs_sub_rois(
            None, sdvol, sblocks.sub_roi_slices, sblocks.sub_rois_offsets, sblocks.denoise_max_shape,
            None, False, [0])
        pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
            sdvol, seg, sblocks.overlap, sblocks.tol, sblocks.sub_roi_slices, sblocks.sub_rois_offsets,
            [0], sblocks.overlap_padding)
        bb = detector.Blobs(pruned)
        bb.replace_rel_with_abs_blob_coords(pruned)
        gpu_final = bb.remove_abs_blob_coords(True)
        parity = bool(cpu_final is not None and gpu_final.shape == cpu_final.shape and
                      np.array_equal(canon(gpu_final), canon(cpu_final)))
        del sdvol

    # ---------------- warm-up, then the timed region
    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    nat.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    final = None
    stats = None
    for _ in range(args.steps):
        final, stats = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = nat.timing_read()
    nat.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        nvox = int(np.prod(shape))
        ns = PROFILE["num_sigma"]
        my_vox = stats.n_voxels                       # block voxels this rank filtered per step
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
        # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
        # command (profiles/, collected with rocprofv3 --pmc in separate passes and calibrated)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_denoise_pmc_traffic.json" if args.denoise
                                   else "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if tuple(shape) == SHAPE and world == 1 and dom in pmc["per_launch_GB"]:
                traffic = round(pmc["per_launch_GB"][dom]["total_GB"] * 1e9)
        except (OSError, KeyError, ValueError):
            pass
        if dom:
            launches = ktimes[dom][1]
            roof = {"bound": "hbm", "kernel": dom, "achieved": stream_k[dom]["alg_GBps"],
                    "note": ("zxpass = fused Z+X pass: its 10 algorithmic B/voxel/sigma replace the 22 of the separate "
                             "Z and X passes, and it is bound by packed-fp32 VALU issue, not by HBM (DESIGN.md "
                             "section 4b); the HBM-bound Y pass of the step runs at "
                             f"{stream_k.get('y2pass', {}).get('alg_GBps', 0) / HBM_PEAK_GBS:.2f} of peak on its "
                             "algorithmic bytes, see 'kernels' and 'pipeline_roofline' for the whole step")
                    if dom == "zxpass" else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(stream_k[dom]["alg_GBps"] / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "alg_bytes_per_launch": int(ALG_BYTES[dom] * my_vox * ns * args.steps / max(1, launches)),
                    "avg_launch_ms": round(ktimes[dom][0] / max(1, launches), 4)}
        gpu_ms = sum(ms for ms, n in ktimes.values()) / args.steps
        out = {
            "metric": "Mvoxels/s, 2048x2048x1024 uint16 stack, 5-sigma LoG blob detection",
            "value": round(nvox * args.steps / elapsed / 1e6, 2), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{shape[2]}x{shape[1]}x{shape[0]} (x,y,z) uint16 Gaussian-blob volume, "
                                   f"seed {SEED}, {n_blocks} blocks (segment_size 256, overlap 5), sigma 3..5 x5, "
                                   "threshold 0.1, overlap 0.5; detect + gather + prune"
                                   + (f"; PLUS per-block preprocessing (denoise_size {args.denoise}: saturate + "
                                      "unsharp + erosion in float64) -- secondary workload, not the headline"
                                      if args.denoise else ""),
                       "blocks_per_rank": hi - lo, "parallelism": f"blocks sharded over {world} GPU(s)"},
            "blobs": 0 if final is None else int(len(final)),
            # digest of the final 8-column table of the last step (computed after the timed region): the same
            # for every path / batch size / rank count that is correct (tests/test_gpu_parity.py compares small
            # volumes with the oracle row by row; this is the full-size cross-check)
            "table_sha1": None if final is None else hashlib.sha1(np.ascontiguousarray(final).tobytes()).hexdigest(),
            "blobs_per_s": round((0 if final is None else len(final)) * args.steps / elapsed, 1),
            "roofline": roof,
            "pipeline_roofline": {
                "alg_bytes_per_voxel": B_ALG_PER_SIGMA * ns,
                "gpu_kernel_ms_per_step_rank0": round(gpu_ms, 2),
                "achieved_GBps_kernels": round(B_ALG_PER_SIGMA * ns * (nvox / world) / (gpu_ms * 1e-3) / 1e9, 1),
                "frac_kernels": round(B_ALG_PER_SIGMA * ns * (nvox / world) / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "achieved_GBps_wall": round(B_ALG_PER_SIGMA * ns * nvox / (elapsed / args.steps) / 1e9, 1),
                "frac_wall": round(B_ALG_PER_SIGMA * ns * nvox / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS / world, 4)},
            "kernels": per_kernel,
            "detector_stats": {k: (round(float(v), 9) if isinstance(v, (float, np.floating)) else int(v))
                               for k, v in vars(stats).items()},
            "cpu_baseline": cpu, "parity_sample_identical": parity,
            "volume_gen_s": round(t_gen, 2),
        }
        print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

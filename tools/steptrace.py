#!/usr/bin/env python3
"""Host timeline of one full-volume step: when each batch is enqueued, when the host starts / ends finishing
it, when the GPU goes idle, how long the tail after the last kernel is and what it consists of.

    python tools/steptrace.py [--config c2|c3|c5] [--denoise 25] [--prune-ahead 0|1] [--fine] [--budget-gb 64]
"""
import argparse, functools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from magellanmapper_amd import blob_log as bl, config, detector, stack_detect, synth

ap = argparse.ArgumentParser()
ap.add_argument("--budget-gb", type=float, default=16.0)
ap.add_argument("--keep-heap", action="store_true", help="mallopt: big arrays from the heap, freed memory stays mapped")
ap.add_argument("--profile", action="store_true", help="cProfile one more step: where the host time goes")
ap.add_argument("--config", default="c3", choices=["c2", "c3", "c5"])
ap.add_argument("--denoise", type=int, default=0, help="profile denoise_size (per-block preprocessing on)")
ap.add_argument("--prune-ahead", default=None, choices=["0", "1"], help="stack_detect.PRUNE_AHEAD")
ap.add_argument("--fine", action="store_true", help="also time the per-block pieces of the co-localisation path")
ap.add_argument("--no-plan", action="store_true", help="no StackDetector.plan_pruning: the whole table is pruned after the detection")
a = ap.parse_args()
if a.keep_heap:
    from magellanmapper_amd import _native
    _native.keep_host_heap()
cfg = bench.CONFIGS[a.config]
shape = cfg["shape"]
dev = torch.device("cuda", 0)
config.resolutions = bench.RESOLUTIONS; config.filename = "p"
config.setup_roi_profiles(None); config.roi_profile.update(dict(bench._BASE_PROFILE, **cfg["profile"]))
if a.prune_ahead is not None:
    stack_detect.PRUNE_AHEAD = a.prune_ahead
if a.denoise:
    config.roi_profile["denoise_size"] = a.denoise
vol = synth.make_volume_device(shape, cfg["seed"], dev)
CHANNELS = list(range(cfg["channels"]))
COLOC = bool(cfg["coloc"])
if len(CHANNELS) > 1:      # (as bench.py builds it: channel 1 = its own blob field + 70 % of channel 0's)
    c1 = synth.make_volume_device(shape, cfg["seed"] + 1, dev)
    c1 = torch.maximum(c1.to(torch.int32), (vol.to(torch.int32) * 7) // 10).to(vol.dtype)
    vol = torch.stack((vol, c1), dim=-1).contiguous()
    del c1
DENOISE = a.denoise or cfg["profile"].get("denoise_size") or 0
config.near_max = [-1.0] * len(CHANNELS)
dvol = bl.DeviceVolume(vol)
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
bl.blob_log_blocks = functools.partial(bl.blob_log_blocks, budget_bytes=int(a.budget_gb * (1 << 30)))
log = []
T0 = [0.0]


def wrap(mod, name, tag):
    fn = getattr(mod, name)

    def inner(*args, **kw):
        t = time.perf_counter()
        out = fn(*args, **kw)
        extra = ""
        if name == "_enqueue_detect":
            extra = f"{out['nb']} blocks"
            ev = torch.cuda.Event(enable_timing=True); ev.record(); out["_ev"] = ev; evs.append(ev)
        log.append((t - T0[0], time.perf_counter() - T0[0], tag, extra))
        return out
    setattr(mod, name, inner)


evs = []
wrap(bl, "_enqueue_detect", "enqueue batch")
if os.environ.get("TRACE_PROLOGUE"):
    wrap(bl, "_make_blocks", "  block records of a batch")
    wrap(bl, "_to_device_bytes", "  block records -> device")
    wrap(bl, "plan_batches", "  plan_batches")
    wrap(bl, "_buffers_for", "  _buffers_for")
    wrap(bl.DeviceVolume, "value_range", "  value_range")
    wrap(bl.DeviceVolume, "value_scale", "  value_scale")
    wrap(bl.ScaleSpace, "device_tables", "  scale-space tables -> device")
    wrap(stack_detect.StackDetector, "plan_pruning", "  plan_pruning")
    wrap(bl._Buffers, "workspace", "  workspace")
    wrap(bl._Buffers, "slots", "  slots")
wrap(bl, "_finish_detect", "finish batch (candidates -> peaks)")
wrap(bl, "_prune_batch", "per-block overlap prune")
wrap(bl, "_prune_batch_native", "per-block overlap prune (native)")
wrap(stack_detect._ArenaSink, "__call__", "tables -> arena (native) + regions that became ready")
wrap(stack_detect._RegionPruner, "__init__", "region pruner set-up")
wrap(stack_detect._RegionPruner, "finish", "remaining regions + merge")
if COLOC and a.fine:
    from magellanmapper_amd import colocalizer
    wrap(detector, "_append_colocs", "  co-localisation of a batch")
    wrap(colocalizer, "colocalize_blocks_device", "    means of one channel (kernel + wait)")
    wrap(colocalizer, "_flags_from_means", "    flags of a block")
    wrap(stack_detect.StackDetector, "_finish_block", "  block table -> ROI coordinates")
    wrap(stack_detect._TableArena, "add", "  table -> arena")
orig_prune = stack_detect.StackPruner.prune_blobs_mp.__func__


def step():
    if not a.no_plan:
        stack_detect.StackDetector.plan_pruning(blocks.overlap, blocks.tol, blocks.overlap_padding, CHANNELS)
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets,
                                                           blocks.denoise_max_shape if DENOISE else None, None,
                                                           COLOC, CHANNELS)
    t = time.perf_counter()
    pruned, _ = stack_detect.StackPruner.prune_blobs_mp(dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                        blocks.sub_rois_offsets, CHANNELS, blocks.overlap_padding,
                                                        final_form=not COLOC, untouched=True)
    log.append((t - T0[0], time.perf_counter() - T0[0], "StackPruner.prune_blobs_mp", ""))
    t = time.perf_counter()
    if isinstance(pruned, stack_detect._FinalTable):
        detector.Blobs(None).cols = list(pruned.col_names); final = pruned.view(np.ndarray)
    else:
        bb = detector.Blobs(pruned); bb.replace_rel_with_abs_blob_coords(pruned); final = bb.remove_abs_blob_coords(True)
    log.append((t - T0[0], time.perf_counter() - T0[0], "final columns", ""))
    return final


for _ in range(2):
    step()
log.clear(); evs.clear()
torch.cuda.synchronize()
start = torch.cuda.Event(enable_timing=True); start.record()
T0[0] = time.perf_counter()
step()
t_end = time.perf_counter() - T0[0]
torch.cuda.synchronize()
gpu_done = [start.elapsed_time(e) for e in evs]
print(f"step wall {t_end * 1e3:.1f} ms; GPU finished the kernels of batch k at {['%.1f' % g for g in gpu_done]} ms")
for t0, t1, tag, extra in log:
    print(f"  {t0 * 1e3:8.2f} -> {t1 * 1e3:8.2f} ms ({(t1 - t0) * 1e3:6.2f})  {tag} {extra}")
print(f"tail after the last kernel: {t_end * 1e3 - gpu_done[-1]:.1f} ms")
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t = time.perf_counter(); step(); ts.append((time.perf_counter() - t) * 1e3)
print("5 more steps (ms):", ["%.1f" % v for v in ts])

if a.profile:
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable(); step(); pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(45)

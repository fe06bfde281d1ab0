#!/usr/bin/env python3
"""Line-ish profile of StackPruner.prune_blobs_mp on a realistic table (needs a GPU for the search)."""
import sys, os, time, inspect, textwrap, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + '/..')
import numpy as np
from magellanmapper_amd import config, stack_detect, stack_prune, detector, roi_prof, chunking
config.resolutions = [[1., 1., 1.]]
shape = (1024, 2048, 2048)
bl = stack_detect.setup_blocks(roi_prof.ROIProfile(segment_size=256, denoise_size=None), shape)
rng = np.random.default_rng(0)
grid = bl.sub_roi_slices.shape
seg = np.zeros(grid, dtype=object)
for c in np.ndindex(*grid):
    sl = bl.sub_roi_slices[c]; lo = np.array([s.start for s in sl]); hi = np.array([s.stop for s in sl])
    n = 1290; t = -np.ones((n, 11)); t[:, :3] = rng.integers(lo, hi, (n, 3)); t[:, 3] = 5.2; t[:, 6] = 0; t[:, 7:10] = t[:, :3]; seg[c] = t
class Img: pass
Img.shape = shape
if "--real" in sys.argv:          # tables (and their arena) from a real detection on the benchmark volume
    import torch
    from magellanmapper_amd import blob_log as _bl, synth as _synth
    import bench as _bench
    config.setup_roi_profiles(None); config.roi_profile.update(_bench._BASE_PROFILE)
    bl = stack_detect.setup_blocks(config.roi_profile, shape)
    _dv = _bl.DeviceVolume(_synth.make_volume_device(shape, 3, torch.device("cuda", 0)))
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, _dv, bl.sub_roi_slices, bl.sub_rois_offsets, None, None, False, [0])
src = inspect.getsource(stack_detect.StackPruner.prune_blobs_mp.__func__)
src = textwrap.dedent(src).replace("@classmethod\n", "")
# insert a timer call after every statement line at the loop-body indentation levels
lines = src.split("\n")
out = []
T = {}
for i, ln in enumerate(lines):
    out.append(ln)
    st = ln.strip()
    ind = len(ln) - len(ln.lstrip())
    if st and not st.startswith(("#", '"""', "def ", "for ", "if ", "else", "elif", "return", "continue", "import")) \
            and not st.endswith((",", "(", "[", "\\", ":")) and ind in (4, 8, 12, 16) and st.count("(") == st.count(")") and '"""' not in st:
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        if len(nxt) - len(nxt.lstrip()) <= ind or not nxt.strip():
            out.append(" " * ind + f"_tick({i})")
code = "\n".join(out)
ns = dict(stack_prune.__dict__)
last = [time.perf_counter()]
def _tick(i):
    now = time.perf_counter(); T[i] = T.get(i, 0) + now - last[0]; last[0] = now
ns["_tick"] = _tick
try:
    exec(code, ns)
except SyntaxError as e:
    print("instrumentation failed", e); sys.exit(0)
fn = ns["prune_blobs_mp"]
for rep in range(3):
    T.clear(); last[0] = time.perf_counter(); t0 = last[0]
    outp, df = fn(stack_detect.StackPruner, Img, seg, bl.overlap, bl.tol, bl.sub_roi_slices, bl.sub_rois_offsets, [0], bl.overlap_padding)
    tot = time.perf_counter() - t0
print("total %.1f ms" % (tot * 1e3), outp.shape)
for i, v in sorted(T.items(), key=lambda e: -e[1])[:14]:
    print("%7.2f ms  %s" % (v * 1e3, lines[i].strip()[:110]))

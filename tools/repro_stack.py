#!/usr/bin/env python3
"""Re-run one dumped mismatch of tools/soak_stack.py (--dump) without replaying the soak's random sequence:
    python tools/repro_stack.py <npz> '<channel-1 profile dict or None>'
MMX_REPRO_ROOT=<checkout> runs another checkout's package (bisecting with git worktrees) against this tree's oracle."""
import ast, os, sys
root = os.environ.get("MMX_REPRO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from magellanmapper_amd import config, preprocess, stack_detect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import magmap_oracle as mmo
preprocess.RGB_GUESS = True
d = np.load(sys.argv[1], allow_pickle=True)
vol, res = d["vol"], d["res"]
over = ast.literal_eval(str(d["over"])); unmix = ast.literal_eval(str(d["unmix"])); coloc = bool(d["coloc"])
ch1 = ast.literal_eval(sys.argv[2]) if len(sys.argv) > 2 else None
nch = vol.shape[3] if vol.ndim == 4 else 1
import tempfile; os.chdir(tempfile.mkdtemp())
config.setup_roi_profiles(["default"] * nch)
for p in config.roi_profiles:
    p.update(over)
if ch1:
    config.roi_profiles[1].update(ch1)
config.roi_profile.update(over)
config.resolutions = res; config.filename = "soak"; config.near_max = [-1.0] * nch
for p in config.roi_profiles:
    p.spectral_unmixing = unmix
config.roi_profile.spectral_unmixing = unmix
profs = [dict(p, spectral_unmixing=unmix) for p in config.roi_profiles]
want, st = mmo.detect_blobs_blocks(vol, None, profs, res, near_max=config.near_max, coloc=coloc)
_, _, blobs = stack_detect.detect_blobs_blocks("soak", stack_detect.Image5d(vol[None]), None, None, None, False, False, True, coloc)
got = blobs.blobs
srt = lambda t: t[np.lexsort(t.T[::-1])]
print("root", root, "got", got.shape, "want", want.shape, "equal", got.shape == want.shape and np.array_equal(srt(got), srt(want)))
print("stats", stack_detect.StackDetector.last_stats)
for c in range(nch):
    print("  channel", c, "got", int((got[:, 6] == c).sum()), "want", int((want[:, 6] == c).sum()))

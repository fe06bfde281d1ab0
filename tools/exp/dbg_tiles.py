import sys, numpy as np
sys.path.insert(0, "/root/repo")
from magellanmapper_amd import blob_log as bl, config, stack_detect, synth, preprocess, _native as nat
from oracle import magmap_oracle as mmo
key = lambda t: t[np.lexsort(tuple(t[:, i] for i in range(t.shape[1] - 1, -1, -1)))]
def run(v):
    img5d = stack_detect.Image5d(v[None])
    _, _, blobs = stack_detect.detect_blobs_blocks("x", img5d, None, None, None, False, False, True, False)
    return blobs.blobs
config.setup_roi_profiles(None)
config.roi_profile.update(dict(num_sigma=3, denoise_size=25, segment_size=40))
config.resolutions = np.array([[1.0, 1.0, 1.0]])
config.filename = "tiles"
for k in range(3):
    v = synth.make_volume(40 + k, (70, 64, 72), 40)
    want, _ = mmo.detect_blobs_blocks(v, None, [dict(config.roi_profile)], config.resolutions)
    for mode in (nat.MMX_PP_SINGLE, nat.MMX_PP_AUTO):
        preprocess.KERNEL_MODE = mode
        got = run(v)
        print(k, mode, got.shape, want.shape, np.array_equal(key(got), key(want)) if got.shape == want.shape else None)

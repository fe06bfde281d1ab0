import sys, numpy as np
sys.path.insert(0, '/root/repo')
from magellanmapper_amd import blob_log as bl, synth, _native as nat
rng = np.random.default_rng(33)
for trial in range(6):
    shape = tuple(int(v) for v in rng.integers(40, 90, 3))
    vol = synth.make_volume(int(rng.integers(1 << 30)), shape, int(rng.integers(150, 500)),
                            blob_sigma=float(rng.uniform(1.2, 2.5)), amp=float(rng.uniform(3000, 30000)))
    if trial % 3 == 2:
        vol = np.tile(vol[:shape[0] // 2, :shape[1] // 2, :shape[2] // 2], (2, 2, 2))
    dvol = bl.DeviceVolume(vol)
    out = []
    for mode in (-1, 7, 6, 2):
        bl.ZX_MODE = mode
        st = bl.BatchStats()
        res, pk = bl.blob_log_blocks(dvol, 0, [(0, 0, 0)], [vol.shape], 1.5, 3.0, 4, 0.02, 0.5, stats=st, return_peaks=True)
        out.append((mode, bl.LAST_ZX_PATH, len(pk[0][1]), st.n_candidates, round(st.max_f32_error, 8)))
    print(trial, vol.shape, out)

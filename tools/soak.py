#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: blob_log (HIP path, default settings) against the oracle on seeded
random volumes -- shapes that span one to several waves per row, every dtype, sigma ranges giving radii 3..24
and the generic path, thresholds, overlaps; single blocks and batches of differently shaped blocks.

    python tools/soak.py [--trials N] [--seed S]
"""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from magellanmapper_amd import blob_log as bl, synth
from oracle import blob_log_oracle as blo

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
bad = 0
rows = 0
t0 = time.time()
for trial in range(a.trials):
    nb = int(rng.choice([1, 1, 2, 4]))
    wide = int(rng.choice([40, 70, 130, 200, 261, 300]))
    full = (int(rng.integers(24, 70)), int(rng.integers(24, 80)), int(wide + rng.integers(0, 12)))
    dt = rng.choice(["u16", "u8", "f32", "f64"], p=[0.55, 0.15, 0.15, 0.15])
    vol = synth.make_volume(int(rng.integers(1 << 30)), full, int(rng.integers(5, 120)),
                            blob_sigma=float(rng.uniform(1.0, 4.0)), amp=float(rng.uniform(3000, 50000)))
    if dt == "u8":
        vol = (vol >> 8).astype(np.uint8)
    elif dt == "f32":
        vol = (vol / 65535.0).astype(np.float32)
    elif dt == "f64":
        vol = vol / 65535.0
    lo = float(rng.uniform(0.8, 4.5))
    hi = lo + float(rng.uniform(0.0, 2.5))
    ns = int(rng.integers(1, 7))
    thr = float(rng.choice([0.02, 0.05, 0.1, 0.2]))
    ov = float(rng.choice([0.0, 0.3, 0.5, 0.9]))
    shapes, origins = [], []
    for _ in range(nb):
        shp = tuple(int(rng.integers(max(8, f // 2), f + 1)) for f in full) if nb > 1 else full
        shapes.append(shp)
        origins.append(tuple(int(rng.integers(0, f - s + 1)) for f, s in zip(full, shp)))
    got = bl.blob_log_blocks(bl.DeviceVolume(vol), 0, origins, shapes, lo, hi, ns, thr, ov)
    for o, shp, g in zip(origins, shapes, got):
        sub = vol[o[0]:o[0] + shp[0], o[1]:o[1] + shp[1], o[2]:o[2] + shp[2]]
        want = blo.blob_log(sub, lo, hi, ns, thr, ov)
        ok = g.shape == want.shape and np.array_equal(g, want)
        rows += len(want)
        if not ok:
            bad += 1
            print("MISMATCH trial", trial, dt, full, shp, o, lo, hi, ns, thr, ov, g.shape, want.shape, flush=True)
print(f"soak seed {a.seed}: {a.trials} trials, {rows} blob rows compared, {bad} mismatching blocks, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

"""Run the tiled Z+X path on the same ragged batch several times: are the candidates' values bit-identical?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import test_gpu_parity as tp
from magellanmapper_amd import _native as nat, synth
vol = synth.make_volume(5, (70, 90, 150), 40)
origins = [(0, 0, 0), (3, 5, 64), (20, 11, 7)]
shapes = [(70, 90, 64), (67, 85, 86), (50, 61, 37)]
for sig in ([3.0, 3.5, 4.0], [1.5, 2.0], [4.5, 5.0]):
    for mode, pre, eps in ((nat.MMX_ZX_TILED, False, 2e-5), (nat.MMX_ZX_AUTO, True, 2.5e-4)):
        ref = None
        bad = 0
        for rep in range(12):
            p, l, k, v, f = tp._abi_batch(vol, origins, shapes, sig, mode, pre, eps=eps)
            if ref is None:
                ref = (k, v)
            elif not (np.array_equal(ref[0], k) and np.array_equal(ref[1], v)):
                bad += 1
                if np.array_equal(ref[0], k):
                    d = np.abs(ref[1] - v)
                    print("   differs:", int((d > 0).sum()), "values, max", float(d.max()), "at", k[np.argmax(d)])
                else:
                    print("   candidate sets differ", len(ref[0]), len(k))
        print(f"sigmas {sig} path {p}: {bad} of 11 repeats differ from the first")

#!/usr/bin/env python3
"""How the oracle pool's C5 sample (bench.py's cpu_baseline leg) scales with the number of processes on this box:
    python tools/exp/cpu_sample_scaling.py 16 32 64      (no GPU is touched)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

if __name__ == "__main__":
    for n in [int(v) for v in sys.argv[1:]] or [16]:
        a = argparse.Namespace(config="c5", denoise=0, segment_size=0, shape=None, cpu_cores=n, cpu_full=False)
        t0 = time.time()
        got = bench.run_cpu_baseline("c5", a, None)
        c = got["cpu"]
        print(json.dumps({"procs": n, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "wall_s": round(time.time() - t0, 1),
                          "detection_s": c["detection_s"], "Mvoxels_per_s": c["value"], "sample": c["sample"][:24]}), flush=True)

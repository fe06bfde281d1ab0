#!/usr/bin/env python3
"""Summarise bench.py JSON lines from stdin."""
import json, sys
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    pr = d["pipeline_roofline"]
    print(f"{' '.join(sys.argv[1:])} value {d['value']} Mvox/s  {d['ms_per_step']} ms/step  gpu_kernels {pr['gpu_kernel_ms_per_step_rank0']} ms"
          f"  frac_kernels {pr['frac_kernels']} frac_wall {pr['frac_wall']}  blobs {d['blobs']}  fallbacks {d['detector_stats']['n_order_fallbacks']}")
    if "-v" in sys.argv:
        print("  roofline", d["roofline"])
        for k, v in d["kernels"].items():
            print("  ", k, v)
        print("  cpu", d.get("cpu_baseline"), "parity", d.get("parity_sample_identical"))

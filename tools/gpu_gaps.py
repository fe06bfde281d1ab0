#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end) over all queues, gaps above
a limit with the kernels either side, and the busy time per family in windows.

    python tools/gpu_gaps.py <dir with *_kernel_trace.csv> [--min-gap-us 300] [--last-ms 300]
"""
import argparse, csv, glob, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--min-gap-us", type=float, default=300.0)
ap.add_argument("--last-ms", type=float, default=300.0, help="look at this much time before the last kernel's end")
a = ap.parse_args()
files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel trace")
rows = []
with open(files[0]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")))
rows.sort()
t_end = max(r[1] for r in rows)
t0 = t_end - int(a.last_ms * 1e6)
rows = [r for r in rows if r[1] >= t0]
busy_until, prev = rows[0][0], rows[0]
idle = 0
print(f"{len(rows)} kernels in the last {a.last_ms} ms")
for r in rows:
    if r[0] > busy_until:
        gap = (r[0] - busy_until) / 1e3
        idle += r[0] - busy_until
        if gap >= a.min_gap_us:
            print(f"  gap {gap:8.1f} us at {(busy_until - t0) / 1e6:8.2f} ms   after {prev[2]!r}  before {r[2]!r}")
    if r[1] > busy_until:
        busy_until, prev = r[1], r
print(f"idle in window: {idle / 1e6:.2f} ms of {(t_end - rows[0][0]) / 1e6:.2f}")

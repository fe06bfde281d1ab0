#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 PMC passes, calibrated on known-byte streams.

usage: pmc_summary.py <dir with trace/ pmc_fetch/ pmc_write/ cal_fetch/ cal_write/>
MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB-like units of the
TCC request counters; on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide (16 B/lane) streams,
WRITE_SIZE is exact for 16 B/lane stores; other widths must be calibrated -- done here with
tools/pmc_calib.py (4 B/lane copy, 16 B/lane copy, u16 read of a known element count).
"""
import csv, glob, os, re, sys, json
from collections import defaultdict

def load_counter(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows

def short(name):
    if any(k in name for k in ("pp_fast_kernel", "pp_generic_kernel", "pp_mid_kernel", "pp_blur_kernel", "pp_stats_kernel",
                               "pp_retile_kernel", "pp_offsets")):
        return "preproc"
    if "zx6_pack" in name:              # (zx6_pack_kernel<T>, zx6_pack_f32_kernel)
        return "zxpack"
    if "zx2_kernel" in name or "zx_kernel" in name or "zx4_kernel" in name:
        return "zxpass"
    if "y2_kernel" in name or "y6_kernel" in name or "ym_kernel" in name:
        return "y2pass"
    if "peaks_sparse_kernel" in name:
        return "peaks"
    m = re.search(r"(zpass|ypass|xpass|peaks|rescore|overlap_pairs|close_pairs|gen_[xyz]|calib_[a-z0-9_]+)_?kernel|(calib_[a-z0-9_]+)", name)
    if not m:
        return None
    return (m.group(1) or m.group(2))

def per_kernel(rows):
    tot, cnt = defaultdict(float), defaultdict(int)
    for name, v in rows:
        k = short(name)
        if k:
            tot[k] += v
            cnt[k] += 1
    return tot, cnt

root = sys.argv[1]
n = 1 << 30
known = {  # bytes per launch of the calibration streams
    "calib_copy_b32": (4 * n, 4 * n), "calib_copy_b128": (4 * n, 4 * n), "calib_read_u16": (2 * n, 0)}
cf, ccf = per_kernel(load_counter(os.path.join(root, "cal_fetch"), "FETCH_SIZE"))
cw, ccw = per_kernel(load_counter(os.path.join(root, "cal_write"), "WRITE_SIZE"))
print("calibration (counter units per launch -> bytes per unit):")
cal = {}
for k, (rb, wb) in known.items():
    fu = cf[k] / max(1, ccf[k]); wu = cw[k] / max(1, ccw[k])
    cal[k] = (rb / fu if fu else float("nan"), wb / wu if wu else float("nan"))
    print(f"  {k:16s} FETCH_SIZE {fu:14.1f} -> {cal[k][0]:8.1f} B/unit   WRITE_SIZE {wu:14.1f} -> {cal[k][1]:8.1f} B/unit")
# access shape -> calibration stream
read_cal = {"zpass": cal["calib_read_u16"][0], "ypass": cal["calib_copy_b32"][0],
            "xpass": cal["calib_copy_b128"][0], "peaks": cal["calib_copy_b128"][0],
            "preproc": cal["calib_read_u16"][0], "zxpass": cal["calib_copy_b128"][0],     # tiled path: 16 B per lane
            "zxpack": cal["calib_read_u16"][0], "y2pass": cal["calib_copy_b32"][0]}
write_cal = {"zpass": cal["calib_copy_b32"][1], "ypass": cal["calib_copy_b32"][1],
             "xpass": cal["calib_copy_b128"][1], "peaks": cal["calib_copy_b128"][1],
             "preproc": cal["calib_copy_b32"][1], "zxpass": cal["calib_copy_b128"][1],
             "zxpack": cal["calib_copy_b128"][1], "y2pass": cal["calib_copy_b32"][1]}
f, fc = per_kernel(load_counter(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"))
w, wc = per_kernel(load_counter(os.path.join(root, "pmc_write"), "WRITE_SIZE"))
res = {}
print("per-kernel HBM traffic per launch (calibrated):")
for k in ("preproc", "zxpack", "zxpass", "y2pass", "zpass", "ypass", "xpass", "peaks"):
    if not fc[k]:
        continue
    rd = f[k] / fc[k] * read_cal[k]
    wr = w[k] / max(1, wc[k]) * write_cal[k]
    res[k] = dict(launches=fc[k], read_GB=rd / 1e9, write_GB=wr / 1e9, total_GB=(rd + wr) / 1e9)
    print(f"  {k:6s} launches {fc[k]:4d}  read {rd/1e9:8.3f} GB  write {wr/1e9:8.3f} GB  total {(rd+wr)/1e9:8.3f} GB")
# ---- issue-side counters of the same command (separate pass): busy fractions of the vector and matrix pipes
# MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, GRBM_GUI_ACTIVE the sum over
# the 8 XCDs; the chip has 256 CUs x 4 SIMDs
busy = {}
sq = {}
for cname in ("SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY"):
    tot, cnt = per_kernel(load_counter(os.path.join(root, "pmc_sq"), cname))
    for k in tot:
        sq.setdefault(k, {})[cname] = tot[k] / max(1, cnt[k])
tcp = {}
for cname in ("TCP_PENDING_STALL_CYCLES", "TCP_TCC_READ_REQ", "TCP_TCC_WRITE_REQ", "SQ_WAVE_CYCLES"):
    tot, cnt = per_kernel(load_counter(os.path.join(root, "pmc_tcp"), cname))
    for k in tot:
        tcp.setdefault(k, {})[cname] = tot[k] / max(1, cnt[k])
print("issue side per launch (fractions of the SIMD cycles of the launch):")
for k, c in sq.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc <= 0:
        continue
    simd = cyc * 1024
    busy[k] = dict(valu_busy=round(4 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd, 4),
                   mfma_busy=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd, 4),
                   cycles_per_xcd=round(cyc))
    t = tcp.get(k, {})
    if t:
        busy[k]["tcp_pending_stall"] = round(t.get("TCP_PENDING_STALL_CYCLES", 0) / (cyc * 256), 4)
        busy[k]["l2_read_requests"] = round(t.get("TCP_TCC_READ_REQ", 0))
        busy[k]["l2_write_requests"] = round(t.get("TCP_TCC_WRITE_REQ", 0))
    print(f"  {k:8s} {busy[k]}")
import hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = {"source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ / TCP, one "
                 "counter set per pass) of python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sub-records; "
                 "HBM bytes calibrated on tools/pmc_calib.py streams; averages over all launches of a kernel family",
       "source_digest": bench.source_digest(), "per_launch_GB": res, "busy": busy}
with open(os.path.join(root, "pmc_counters.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))

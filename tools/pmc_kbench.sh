#!/bin/bash
# rocprofv3 PMC passes over tools/kbench.py (one kernel family at benchmark geometry)
#   usage: pmc_kbench.sh <tag> <fuse mode> "<counter set 1>" "<counter set 2>" ...
set -u
tag=$1; export MMX_FUSE=$2; shift 2
out=gpurun_out/pmck_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python tools/kbench.py --blocks 16 --sigmas 4 --reps 1 --mask > $out/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $out/p$i.log)"
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = ("pack" if "pack_kernel" in k else "zx" if ("zx" in k and "setup" not in k) else
             "y2" if "y2_kernel" in k else "y6" if "y6_kernel" in k else "ym" if "ym_kernel" in k else None)
        if k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in tot:
    for c in sorted(tot[k]):
        print(f"{k:4s} {c:36s} per launch {tot[k][c] / cnt[k][c]:16.1f}  (launches {cnt[k][c]})")
PY
rm -rf $out/p*/

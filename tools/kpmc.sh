#!/bin/bash
# PMC counters of one command on the GPU box (run through gpurun); counters in their own pass, kernel-trace only:
#   tools/kpmc.sh <tag> "<counter list>" python3 tools/ppbench.py ...
# Prints per kernel name the mean of every counter over its dispatches; CSV under gpurun_out/kpmc_<tag>/.
set -u
tag=$1; shift
ctrs=$1; shift
out=gpurun_out/kpmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- "$@" > $out/run.log 2>&1
python3 - "$out" "${FILTER:-}" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    sys.exit("no counter_collection.csv: " + open(sys.argv[1] + "/run.log").read()[-600:])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] and sys.argv[2] not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-32s n=%-4d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
find $out -name "*counter_collection.csv" -size +4M -delete
find $out -name "*kernel_trace.csv" -size +4M -delete
find $out -name "*.db" -delete

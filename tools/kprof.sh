#!/bin/bash
# Per-kernel durations of one command on the GPU box (run through gpurun):
#   tools/kprof.sh <tag> python3 tools/ppbench.py ...      (the program itself after the tag: no env / bash -c hops)
# Prints the kernel-stats table (name, calls, total ms, average us) and leaves the CSV under gpurun_out/kprof_<tag>/.
set -u
tag=$1; shift
out=gpurun_out/kprof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $out/run.log 2>&1
tail -2 $out/run.log
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    sys.exit("no kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    print("%-90s %6s %10.3f ms %10.2f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $out -name "*kernel_trace.csv" -size +4M -delete
find $out -name "*.db" -delete

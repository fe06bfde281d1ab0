#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   kernel trace + stats of bench.py, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the
#   bench and for the known-byte calibration streams.  Outputs under gpurun_out/prof_<tag>/.
set -u
tag=${1:-r01}
extra=${2:-}          # extra bench.py arguments, e.g. "--denoise 25"
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline $extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $B > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $B > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $B > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- python tools/pmc_calib.py > $out/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- python tools/pmc_calib.py > $out/cal_write.log 2>&1
python tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
# keep the merge small
find $out -name "*kernel_trace.csv" -size +8M -delete
find $out -name "*counter_collection.csv" -size +8M -delete

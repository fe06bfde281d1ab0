#!/bin/bash
# The LoG kernels on PREPROCESSED (float) voxels, alone: bench.py --denoise 25 with the preprocessing on the LoG stream
# (--pre-stream 0: the kernel families run one after the other, so every family's HIP-event time is its own), beside
# the raw-uint16 run of the same volume; FETCH_SIZE / WRITE_SIZE passes of the float run for its traffic.
#   tools/logfloat_profile.sh [tag]   -> gpurun_out/logfloat_<tag>/
set -u
tag=${1:-r06}
out=gpurun_out/logfloat_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sub-records"
timeout 300 python $B > $out/raw.json 2> $out/raw.err
timeout 300 python $B --denoise 25 --pre-stream 0 > $out/float_alone.json 2> $out/float_alone.err
timeout 300 python $B --denoise 25 > $out/float_two_streams.json 2> $out/float_two_streams.err
P="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sub-records --denoise 25 --pre-stream 0"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $P > $out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $P > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- python tools/pmc_calib.py > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- python tools/pmc_calib.py > $out/cal_write.log 2>&1
python tools/pmc_summary.py $out > $out/pmc_summary.txt 2>&1
python - <<PY
import json
def k(f):
    r = json.load(open(f)); return r["ms_per_step"], {n: (v["ms_per_step"], v["launches_per_step"]) for n, v in r["kernels"].items()}, r["table_sha1"]
raw, alone, two = k("$out/raw.json"), k("$out/float_alone.json"), k("$out/float_two_streams.json")
print("step ms: raw %.2f, --denoise 25 one stream %.2f, two streams %.2f" % (raw[0], alone[0], two[0]))
print("%-8s %10s %14s %14s" % ("kernel", "raw u16", "float (alone)", "float (beside preproc)"))
for n in ("zxpack", "zxpass", "y2pass", "peaks", "rescore", "preproc"):
    print("%-8s %10s %14s %14s" % (n, raw[1].get(n, ("-",))[0], alone[1].get(n, ("-",))[0], two[1].get(n, ("-",))[0]))
PY
tail -12 $out/pmc_summary.txt
find $out -name "*kernel_trace.csv" -size +4M -delete
find $out -name "*counter_collection.csv" -size +4M -delete
find $out -name "*.db" -delete

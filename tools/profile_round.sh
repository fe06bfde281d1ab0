#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun): bench lines of the three configs, the
# rocprofv3 kernel trace + stats of the headline command, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE)
# for the bench and for the known-byte calibration streams.  Outputs under gpurun_out/prof_<tag>/.
set -u
tag=${1:-r06}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the driver's command (c3 with the compact c2 / c5 sub-records and the CPU baselines), then the configs on their own
timeout 900 python bench.py --steps 10 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err
timeout 300 python bench.py --config c2 --steps 40 --warmup 5 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c5.json 2> $out/bench_c5.err
timeout 600 python bench.py --denoise 25 --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_denoise25.json 2> $out/bench_denoise25.err
timeout 600 python bench.py --config c5 --tiles 2 --steps 4 --warmup 1 --parity-sample tests/golden/bench_sample_c5.npz > $out/bench_c5_tiles2.json 2> $out/bench_c5_tiles2.err
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sub-records"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $B > $out/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $B > $out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $B > $out/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- python $B > $out/pmc_sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ SQ_WAVE_CYCLES --output-format csv -d $out/pmc_tcp -- python $B > $out/pmc_tcp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- python tools/pmc_calib.py > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- python tools/pmc_calib.py > $out/cal_write.log 2>&1
python tools/pmc_summary.py $out > $out/summary.txt 2>&1
tail -30 $out/summary.txt
# keep the merge small
find $out -name "*kernel_trace.csv" -size +4M -delete
find $out -name "*counter_collection.csv" -size +4M -delete
find $out -name "*.db" -delete
du -sh $out

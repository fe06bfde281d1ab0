mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
sum() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels']
    print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'sha', (d['table_sha1'] or '')[:8], {a:b['ms_per_step'] for a,b in k.items()}, 'exposed', d['pipeline_roofline']['host_exposed_ms_per_step'], 'parity', d['parity_sample_identical'], 'graph', d.get('graph_replay'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "q16_error_bound_holds or native_call or replay or through_the_abi or band_narrower" 2>&1 | tail -60 > $O/pytest_new.txt; tail -40 $O/pytest_new.txt
python bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline > $O/c2.json 2> $O/c2.err; sum $O/c2.json
MMX_GRAPH_BLOCKS=0 python bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_nograph.json 2> $O/c2_nograph.err; sum $O/c2_nograph.json
MMX_NATIVE_BATCH=0 python bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_old.json 2> $O/c2_old.err; sum $O/c2_old.json
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --parity-sample tests/golden/bench_sample_c5.npz > $O/c5.json 2> $O/c5.err; sum $O/c5.json
MMX_PRE_AHEAD=1 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5_ahead1.json 2> $O/c5_ahead1.err; sum $O/c5_ahead1.json
python bench.py --denoise 25 --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records > $O/den25.json 2> $O/den25.err; sum $O/den25.json
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records > $O/c3.json 2> $O/c3.err; sum $O/c3.json
MMX_NATIVE_BATCH=0 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records > $O/c3_old.json 2> $O/c3_old.err; sum $O/c3_old.json
# counters: one tile per wave vs two (16 blocks, sigma 4 = radius 16, one launch each)
MMX_ZX_PAIR=0 bash tools/pmc_kbench.sh pair0 7 "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY SQ_BUSY_CYCLES" > $O/pmc_pair0.txt 2>&1
MMX_ZX_PAIR=1 bash tools/pmc_kbench.sh pair1 7 "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY SQ_BUSY_CYCLES" > $O/pmc_pair1.txt 2>&1
grep "^zx" $O/pmc_pair0.txt; echo; grep "^zx" $O/pmc_pair1.txt

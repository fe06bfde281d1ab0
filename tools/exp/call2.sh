mkdir -p gpurun_out/r04b
B="python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records"
sum() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels']
    print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'sha', d['table_sha1'][:8], 'zx', k['zxpass']['ms_per_step'], 'y', k['y2pass']['ms_per_step'], 'maxerr', d['detector_stats']['max_f32_error'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
# correctness of the pair kernel first: the parity tests that sweep radii and full pipeline
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "radius or q16 or blob_log_identical or tiled or two_blocks" 2>&1 | tail -5 > gpurun_out/r04b/pytest_pair.txt; cat gpurun_out/r04b/pytest_pair.txt
MMX_ZX_PAIR=1 $B > gpurun_out/r04b/pair1.json 2> gpurun_out/r04b/pair1.err; sum gpurun_out/r04b/pair1.json
MMX_ZX_PAIR=0 $B > gpurun_out/r04b/pair0.json 2> gpurun_out/r04b/pair0.err; sum gpurun_out/r04b/pair0.json
MMX_ZX_PAIR=1 $B > gpurun_out/r04b/pair1b.json 2> gpurun_out/r04b/pair1b.err; sum gpurun_out/r04b/pair1b.json
for n in 3 5 8; do MMX_ZX_PAIR=0 MMX_SUBBATCH=$n $B > gpurun_out/r04b/sub$n.json 2> gpurun_out/r04b/sub$n.err; sum gpurun_out/r04b/sub$n.json; done
MMX_ZX_PAIR=1 MMX_SUBBATCH=5 $B > gpurun_out/r04b/pair1sub5.json 2> gpurun_out/r04b/pair1sub5.err; sum gpurun_out/r04b/pair1sub5.json
python tools/benchprof.py --config c2 --steps 300 --warmup 20 > gpurun_out/r04b/prof_c2.json 2> gpurun_out/r04b/prof_c2.txt
python tools/benchprof.py --config c5 --steps 2 --warmup 1 > gpurun_out/r04b/prof_c5.json 2> gpurun_out/r04b/prof_c5.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04b/pytest_all.txt; cat gpurun_out/r04b/pytest_all.txt

O=gpurun_out/r04j; mkdir -p $O
J() { python - "$@" <<'PY'
import json,sys
for path in sys.argv[1:]:
    for l in open(path):
        if l.startswith('{'):
            d=json.loads(l)
            print(path.split('/')[-1], 'ms', d['ms_per_step'], 'sha', (d['table_sha1'] or '')[:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()}, 'exposed', d['pipeline_roofline']['host_exposed_ms_per_step'], 'prewait', d['pipeline_roofline'].get('pre_stream_wait_ms'), 'parity', d['parity_sample_identical'])
PY
}
timeout 1200 python -m pytest tests/test_gpu_preproc.py tests/test_gpu_coloc.py -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "denoise or stack or soak or plateau" 2>&1 | tail -3
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --parity-sample tests/golden/bench_sample_c5.npz > $O/c5_side.json 2> $O/c5_side.err; J $O/c5_side.json
MMX_PRE_SIDE_TAIL=0 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5_main.json 2> $O/c5_main.err; J $O/c5_main.json
python bench.py --denoise 25 --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records > $O/den_side.json 2> $O/den_side.err; J $O/den_side.json
MMX_PRE_SIDE_TAIL=0 python bench.py --denoise 25 --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records > $O/den_main.json 2> $O/den_main.err; J $O/den_main.json
python __graft_entry__.py smoke 2>&1 | tail -3

mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "replay or tiled_path_entries or prepacked or two_blocks or radius" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_configs.py -q -k "two_blocks or c2_single" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records > $O/c3_$i.json 2> $O/c3_$i.err; python - $O/c3_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d['ms_per_step'], d['table_sha1'][:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()}, d['pipeline_roofline']['host_exposed_ms_per_step'])
PY
done
python bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/c2.json 2> $O/c2.err; python - $O/c2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('c2', d['ms_per_step'], d['table_sha1'][:8], d.get('graph_replay'), d['pipeline_roofline']['host_exposed_ms_per_step'])
PY

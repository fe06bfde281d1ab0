O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rescore or blob_log_identical or plateau or through_the_abi or native_call or band" 2>&1 | tail -4
B="python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records"
for i in 1 2; do $B > $O/c3_$i.json 2> $O/c3_$i.err; python - $O/c3_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('c3', d['ms_per_step'], d['table_sha1'][:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()}, d['pipeline_roofline']['host_exposed_ms_per_step'])
PY
done
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5.err; python - $O/c5.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('c5', d['ms_per_step'], d['table_sha1'][:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()})
PY

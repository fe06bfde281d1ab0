mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
J() { python - "$@" <<'PY'
import json,sys
for path in sys.argv[1:]:
    for l in open(path):
        if l.startswith('{'):
            d=json.loads(l)
            print(path.split('/')[-1], 'ms', d['ms_per_step'], 'sha', (d['table_sha1'] or '')[:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()}, 'exposed', d['pipeline_roofline']['host_exposed_ms_per_step'], 'replays', d.get('graph_replay'), 'ranks', d.get('ranks'))
PY
}
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_all.txt; cat $O/pytest_all.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records > $O/c3.json 2> $O/c3.err; J $O/c3.json
python bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/c2.json 2> $O/c2.err; J $O/c2.json
MMX_PRUNE_PROF=1 MMX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/ranks4.json 2> $O/ranks4.err; J $O/ranks4.json
grep "distributed prune" $O/ranks4.err | tail -9
python tools/steptrace.py --keep-heap > $O/steptrace_c3.txt 2>&1; sed -n 1,6p $O/steptrace_c3.txt; tail -8 $O/steptrace_c3.txt

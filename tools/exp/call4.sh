mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_all.txt; cat $O/pytest_all.txt
python bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d/c2.json')); print('c2', d['ms_per_step'], d['value'], 'replays', d['graph_replay'], d['kernels'].keys(), d['pipeline_roofline']['host_exposed_ms_per_step'])
PY
MMX_GRAPH_BLOCKS=0 python bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/c2_nograph.json 2> $O/c2_nograph.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d/c2_nograph.json')); print('c2 nograph', d['ms_per_step'], d['value'], 'replays', d['graph_replay'])
PY
python tools/steptrace.py --keep-heap > $O/steptrace_c3.txt 2>&1; tail -45 $O/steptrace_c3.txt

mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "q16_error_bound_holds" 2>&1 | tail -3
MMX_PRUNE_PROF=1 MMX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/ranks4.json 2> $O/ranks4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/ranks4.json')); print('4 ranks', d['ms_per_step'], d['table_sha1'][:8], d['ranks'])
PY
grep "distributed prune" $O/ranks4.err | tail -12
MMX_PRUNE_PROF=1 MMX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/ranks2.json 2> $O/ranks2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/ranks2.json')); print('2 ranks', d['ms_per_step'], d['table_sha1'][:8], d['ranks'])
PY
grep "distributed prune" $O/ranks2.err | tail -12
TRACE_PROLOGUE=1 python tools/steptrace.py --keep-heap > $O/steptrace_prologue.txt 2>&1; sed -n 1,40p $O/steptrace_prologue.txt
python tools/benchprof.py --config c2 --steps 300 --warmup 20 --top 70 > $O/prof_c2.json 2> $O/prof_c2.txt
grep -A75 "Ordered by: cumulative" $O/prof_c2.txt | grep -v "importlib\|<module>\|__import__\|_bootstrap" | head -75

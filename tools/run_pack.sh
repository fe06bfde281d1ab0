timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for b in 16 24; do
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --budget-gb $b 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print($b, d['ms_per_step'], d['value'], d['table_sha1'][:8], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done

timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['detector_stats']; print(d['ms_per_step'], d['value'], d['table_sha1'][:8], {k:s[k] for k in ('n_candidates','n_contested','n_probes','n_band_retries','max_f32_error')}, {k:v['ms_per_step'] for k,v in d['kernels'].items()})"

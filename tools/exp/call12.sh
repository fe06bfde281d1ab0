O=gpurun_out/r04l; mkdir -p $O
B="python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records"
for v in 1 0 1 0; do MMX_YM_REVERSE=$v $B > $O/rev$v.json 2> $O/rev$v.err; python - $O/rev$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('reverse', sys.argv[2], d['ms_per_step'], d['table_sha1'][:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()})
PY
done

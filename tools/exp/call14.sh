O=gpurun_out/r04n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "radius or tiled or q16 or blob_log_identical or block_shape or float_voxels or y_pass" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_configs.py -q -k "two_blocks or c2_single" 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records"
for v in 0 1 0 1; do MMX_ZX_KEEP_PAD=$v $B > $O/pad$v.json 2> $O/pad$v.err; python - $O/pad$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('keep_pad', sys.argv[2], d['ms_per_step'], d['table_sha1'][:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()})
PY
done

O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "radius or tiled or q16 or blob_log_identical or float_voxels" 2>&1 | tail -3
for i in 1 2; do
for lib in stock noze; do
  if [ $lib = noze ]; then export MMX_LIB_PATH=$PWD/magellanmapper_amd/libmmx_noze.so; else unset MMX_LIB_PATH; fi
  echo "== $lib: kbench 64 blocks, sigma 3 4 4.5 5 (R 12 16 18 20), MMX_FUSE=7"
  MMX_FUSE=7 python tools/kbench.py --blocks 64 --sigmas 3 4 4.5 5 --reps 3 --mask 2>&1 | grep -E "^zx|zxpass" | head -8
done; done
unset MMX_LIB_PATH
B="python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records"
for lib in stock noze stock noze; do
  if [ $lib = noze ]; then export MMX_LIB_PATH=$PWD/magellanmapper_amd/libmmx_noze.so; else unset MMX_LIB_PATH; fi
  $B > $O/$lib.json 2> $O/$lib.err; python - $O/$lib.json $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['ms_per_step'], d['table_sha1'][:8], {a:b['ms_per_step'] for a,b in d['kernels'].items()})
PY
done

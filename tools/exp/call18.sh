export MMX_DIST_BACKEND=gloo
( time python bench.py --gpus 4 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/a.json 2> /tmp/a.err ) 2>&1 | grep real
( time python bench.py --gpus 4 --steps 1 --warmup 0 --no-cpu-baseline --parity-sample tests/golden/bench_sample_c3.npz > /tmp/b.json 2> /tmp/b.err ) 2>&1 | grep real
tail -3 /tmp/b.err
python - <<'PY'
import time, sys
sys.path.insert(0,'.')
import bench, numpy as np
t=time.time(); s=bench.make_host_sample((320,512,512), 3, 1); print('make_host_sample c3', round(time.time()-t,1), 's')
t=time.time(); s=bench.make_host_sample((96,512,512), 3, 2); print('make_host_sample c5', round(time.time()-t,1), 's')
PY

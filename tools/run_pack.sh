MMX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 tools/soak_ranks.py --trials 25 --seed 34 2>&1 | tail -2 | tee -a gpurun_out/soak_r02_q16c.txt
for n in 2 4; do
MMX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_ranks$n.json
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_ranks$n.json').read())
print($n, d['ms_per_step'], d['value'], d['table_sha1'][:8], d['ranks'])
PY
done

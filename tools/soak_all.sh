#!/bin/bash
# The randomised parity soaks back to back (GPU box): usage tools/soak_all.sh <seed> <scale>
s=${1:-1}; k=${2:-1}
timeout 3000 python tools/soak.py --trials $((1500 * k)) --seed $((s + 1)) 2>&1 | tail -1
timeout 3000 python tools/soak_stack.py --trials $((400 * k)) --seed $((s + 2)) 2>&1 | tail -1
timeout 1500 python tools/soak_detect.py --trials $((500 * k)) --seed $((s + 3)) 2>&1 | tail -1
timeout 1500 python tools/soak_preproc.py --trials $((150 * k)) --seed $((s + 4)) 2>&1 | tail -1
MMX_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 tools/soak_ranks.py --trials $((40 * k)) --seed $((s + 5)) 2>&1 | tail -1

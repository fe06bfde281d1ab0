"""bench.py --gpus 2 with the pruning phases printed per step (stack_prune.PRUNE_PROF):

    MMX_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/exp/ranks_prof.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
from magellanmapper_amd import stack_prune
stack_prune.PRUNE_PROF = True
import bench
bench.main()

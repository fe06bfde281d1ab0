import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magellanmapper_amd import synth
r = int(sys.argv[1]); n = int(sys.argv[2])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
nz = 1024
z0, z1 = r * nz // n, (r + 1) * nz // n + (5 if r < n - 1 else 0)
t = time.time()
v = synth.make_volume_device((1024, 2048, 2048), 3, dev, z_range=(z0, z1))
torch.cuda.synchronize()
print(f"rank {r}/{n}: slab {z0}..{z1} generated in {time.time() - t:.2f} s, sum {int(v.to(torch.int64).sum())}", flush=True)

import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from magellanmapper_amd import blob_log as bl, synth
import bench
dev=torch.device('cuda',0)
vol=synth.make_volume_device((261,261,261),3,dev)   # same generator/density as the benchmark
dvol=bl.DeviceVolume(vol)
space=bl.ScaleSpace.make(3,5,5)
cube=bl.log_cube_blocks(dvol,0,[(0,0,0)],[(261,261,261)],space)[0]   # (ns? ) 
print(cube.shape)
c=np.moveaxis(cube,-1,0) if cube.shape[-1]==5 else cube
thr=0.1-2e-5
for s in range(c.shape[0]):
    a=c[s]                       # z,y,x
    px=288
    pad=np.full((261,261,px),-1.0,np.float32); pad[:,:,:261]=a
    # columns col=z*px+x for each y: segments of 64 along (z,x) flattened
    m=np.moveaxis(pad,1,0).reshape(261,-1)   # y, z*px+x
    n=m.shape[1]//64*64
    seg=(m[:,:n].reshape(261,-1,64)>thr).any(-1)
    vox=(a>thr).mean()
    print('sigma',s,'voxels above %.4f'%vox,'live 64-col segments %.4f'%seg.mean())

"""K optimisation steps of a fitting_video window (fit type 1234, 4 frames x 40 rays, stable loss) for rocprofv3."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda')
renb, netsb, chainb, viewsb, ov = bench.build_fit(dev, 60, bench.VID_FRAMES, bench.VID_RAYS, 'f16x3', halo=(sys.argv[2] if len(sys.argv) > 2 else 'halo') == 'halo')
optb = F.make_optimizer(chainb, video=True)
idx = list(range(bench.VID_FRAMES))
def step(i):
    F.fit_step(renb, viewsb[i % 8], chainb, optb, bench.NEAR, bench.FAR, '1234', index=idx, smooth_ends=(True, False), obj_verts_for_stable=ov)
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i)
torch.cuda.synchronize()
print('ms per window step: %.3f' % ((time.perf_counter() - t0) / steps * 1e3))

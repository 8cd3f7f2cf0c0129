"""K optimisation steps of fitting_single (fit type 12, 196 rays x 192 depths, both fields) for rocprofv3:
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fit -- python3 tools/fit_profile.py [steps] [halo|rigid] [pipe|autograd] [frames]
   frames > 1: that many independent frames side by side through the same launches (fitting.fit_frames_batched's step)."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda')
halo = (sys.argv[2] if len(sys.argv) > 2 else 'halo') == 'halo'     # the reference's six-leaf pose chain (default) or the rigid one
ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=halo)
opt = F.make_optimizer(chain, video=False)
pipe = (sys.argv[3] if len(sys.argv) > 3 else 'pipe') == 'pipe'     # the two-stream step (fit_frame's default) or the autograd step
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if frames > 1:
    made = []
    for f in range(frames):
        ch, jf, _ = bench.build_fit_data(dev, 140 + f, 1, halo=True)
        made.append((F.synthetic_views(8, 1, bench.FIT_RAYS, 140 + f, jf[9], device=dev), ch))
    stacked = F.HaloPoseChain.stack([m[1] for m in made])
    opt = F.make_optimizer(stacked, video=False)
    fit = F.PipelinedSingleFit(ren, stacked, opt, bench.NEAR, bench.FAR, '12')
    bviews = [F.stack_views([m[0][v] for m in made]) for v in range(8)]
    for i in range(3):
        fit.step(bviews[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fit.step(bviews[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print('ms per step: %.3f  (%d frames side by side: %.3f ms per frame-step)' % (dt, frames, dt / frames))
    sys.exit(0)
for i in range(3):
    F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12', pipelined=pipe)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12', pipelined=pipe)
torch.cuda.synchronize()
print('ms per step: %.3f' % ((time.perf_counter() - t0) / steps * 1e3))

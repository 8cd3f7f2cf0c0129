"""Host (enqueue) time per step against the step's period: is a fitting loop bound by the host issuing its launches or by the GPU?
For each form: K steps timed twice -- the host's time to ISSUE them (no synchronisation inside; one at the end, not counted) and the
wall time including the final synchronisation."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from honerf_amd import fitting as F  # noqa: E402

dev = torch.device('cuda')
K = 60


def measure(name, step):
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-44s host issue %.3f ms / step, period %.3f ms / step' % (name, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3), flush=True)


ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True)
opt = F.make_optimizer(chain, video=False)
measure('fitting_single, pipelined (1 frame)', lambda i: F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12', pipelined=True))
F.finish_pipeline(opt)
measure('fitting_single, autograd (1 frame)', lambda i: F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12'))
for frames in (4, 8):
    made = []
    for f in range(frames):
        ch, jf, _ = bench.build_fit_data(dev, 140 + f, 1, halo=True)
        made.append((F.synthetic_views(8, 1, bench.FIT_RAYS, 140 + f, jf[9], device=dev), ch))
    stacked = F.HaloPoseChain.stack([m[1] for m in made])
    o2 = F.make_optimizer(stacked, video=False)
    fit = F.PipelinedSingleFit(ren, stacked, o2, bench.NEAR, bench.FAR, '12')
    bviews = [F.stack_views([m[0][v] for m in made]) for v in range(8)]
    measure('fitting_single, pipelined (%d frames side by side)' % frames, lambda i: fit.step(bviews[i % 8]))
    fit.finish()
renb, netsb, chainb, viewsb, verts = bench.build_fit(dev, 41, 4, bench.VID_RAYS, 'f16x3', halo=True)
optb = F.make_optimizer(chainb, video=True)
ov = verts.contiguous()
measure('fitting_video window step (autograd)', lambda i: F.fit_step(renb, viewsb[i % 8], chainb, optb, bench.NEAR, bench.FAR, '1234', index=[0, 1, 2, 3],
                                                                     smooth_ends=(True, False), obj_verts_for_stable=ov))

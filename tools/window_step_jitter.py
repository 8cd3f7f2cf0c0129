"""The fitting_video window step in blocks of 20 steps (a synchronisation per block): ms per step of every block, and host time per step
of the slow ones -- where the jitter of `video_1234_step` comes from.  python tools/window_step_jitter.py [blocks]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda')
renb, netsb, chainb, viewsb, ov = bench.build_fit(dev, 60, bench.VID_FRAMES, bench.VID_RAYS, 'f16x3', halo=True)
optb = F.make_optimizer(chainb, video=True)
idx = list(range(bench.VID_FRAMES))
def step(i):
    F.fit_step(renb, viewsb[i % 8], chainb, optb, bench.NEAR, bench.FAR, '1234', index=idx, smooth_ends=(True, False), obj_verts_for_stable=ov)
for i in range(10): step(i)
torch.cuda.synchronize()
res = []
for b in range(blocks):
    t0 = time.perf_counter()
    host = []
    for i in range(20):
        h0 = time.perf_counter()
        step(b * 20 + i)
        host.append((time.perf_counter() - h0) * 1e3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.append(((t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, max(host)))
ms = sorted(r[0] for r in res)
print('blocks of 20 steps: median %.3f ms per step, min %.3f, max %.3f' % (ms[len(ms) // 2], ms[0], ms[-1]))
for b, (total, issue, worst) in enumerate(res):
    flag = '  <--' if total > 1.15 * ms[len(ms) // 2] else ''
    print('block %2d: %.3f ms per step; host issue %.3f ms per step, slowest single issue %.2f ms%s' % (b, total, issue, worst, flag))

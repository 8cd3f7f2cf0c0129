"""Does the tile-level early-out (hn_field_set_culling) compose with the sample-level far-field skip in a fitting step?
Same step with culling off / on: results and time.  python tools/cull_in_fit_probe.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F
dev = torch.device('cuda')
res = {}
for cull in (False, True, False, True):
    ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True)
    ren.fields()[0].set_culling(cull)
    opt = F.make_optimizer(chain, video=False)
    tr = torch.rand(bench.FIT_RAYS, 1, generator=torch.Generator().manual_seed(3)).to(dev)
    terms = F.fit_backward(ren, views[0], chain, bench.NEAR, bench.FAR, '12', t_rand=tr)
    torch.cuda.synchronize()
    res[cull] = ({k: float(v.detach()) for k, v in terms.items()}, [p.grad.clone() for p in chain.parameters()])
    for i in range(5):
        F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12')
    torch.cuda.synchronize()
    print('culling %s: %.3f ms per step' % (cull, (time.perf_counter() - t0) / 60 * 1e3), flush=True)
for k in res[False][0]:
    print(k, res[False][0][k], res[True][0][k])
for a, b in zip(res[False][1], res[True][1]):
    print('grad max abs diff %.3e  (max abs %.3e)' % (float((a - b).abs().max()), float(a.abs().max())))

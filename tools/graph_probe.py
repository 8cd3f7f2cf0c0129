"""Can a whole fitting_single step be captured in a HIP graph and replayed?  Times eager vs replay."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F

dev = torch.device('cuda')
ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3')
opt = torch.optim.Adam(chain.param_groups(video=False), capturable=True)
static = {k: (v.clone() if isinstance(v, torch.Tensor) else {kk: vv.clone() for kk, vv in v.items()}) for k, v in views[0].items()}
t_rand = torch.rand(bench.FIT_RAYS, 1, device=dev)

def step():
    return F.fit_step(ren, static, chain, opt, bench.NEAR, bench.FAR, '12', t_rand=t_rand)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print('eager ms/step %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    terms = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    for k in ('xy', 'true_rgb', 'true_mask'):
        static[k].copy_(views[i % 8][k])
    t_rand.uniform_()
    g.replay()
torch.cuda.synchronize()
print('graph ms/step %.3f  loss %.5f' % ((time.perf_counter() - t0) / 20 * 1e3, float(terms['loss'])))

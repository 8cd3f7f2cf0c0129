"""Host time to ISSUE a training iteration against the iteration's period (tools/train_step_bench.py's batch): if the two are close the
loop is bound by the host (Python + launches) or the host is waiting for the device somewhere, not running ahead of it.  (Round 5: the
object loop issued in 4.15 of 4.20 ms while every re-pack read the trained variance back; 2.2 of 3.9 ms since it stays on the device.)      python tools/train_host_probe.py <obj|hand>"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
import torch
import train_step_bench as B
from honerf_amd import training

kind = sys.argv[1] if len(sys.argv) > 1 else 'obj'
dev = torch.device('cuda:0')
ren, synth = B.build(kind, dev)
ren.precision = 'f16x3'
n_rays = 441
o, d, ex = B.rays(kind, synth, n_rays, dev)
g = torch.Generator(device='cpu').manual_seed(5)
true_rgb = torch.rand(n_rays, 3, generator=g).to(dev)
true_mask = (torch.rand(n_rays, 1, generator=g) > 0.3).float().to(dev)
opt = training.make_optimizer(ren, 1e-4)


def step(marks=None):
    t = [time.perf_counter()]
    ren.mark_parameters_changed()
    ren.field()
    t.append(time.perf_counter())
    out = training.render_train(ren, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], None, ex['Ro'], ex['To'], repack=False)
    terms = training.train_loss(out, true_rgb, true_mask, 1.0, 1.0)
    t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True)
    terms['loss'].backward()
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    if marks is not None:
        for i in range(4):
            marks[i] += t[i + 1] - t[i]


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 40
marks = [0.0] * 4
t0 = time.perf_counter()
for _ in range(n):
    step(marks)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('%s: host issues an iteration in %.2f ms (re-pack %.2f, render + loss %.2f, backward %.2f, optimiser %.2f); the iteration\'s period is %.2f ms' % (
    kind, t_issue / n * 1e3, marks[0] / n * 1e3, marks[1] / n * 1e3, marks[2] / n * 1e3, marks[3] / n * 1e3, t_all / n * 1e3))
# the re-pack alone, the GPU idle before each call: its own host cost (anything above it in the loop is waiting for the device)
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ren.mark_parameters_changed()
    ren.field()
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print('%s: re-pack with the GPU idle: host %.2f ms (min %.2f)' % (kind, sum(ts) / len(ts) * 1e3, min(ts) * 1e3))
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    ren.mark_parameters_changed()
    ren.field()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
print(s.getvalue()[:3500])

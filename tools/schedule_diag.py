"""Diagnostic: fit_sequence_video (1 rank) against fit_step in the reference order, each run twice, pairwise parameter distances."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
from honerf_amd import fitting as F
from test_gpu_surface import _dual
dev = torch.device('cuda')
data_num, n_views, outer, sub = 6, 2, 2, 1
renb = _dual(True)
def problem():
    chain, j, verts = bench.build_fit_data(dev, 60, data_num, halo=True, drift=0.002)
    per_window = {tuple(w): F.synthetic_views(n_views, 4, 6, 300 + w[0], j[9], device=dev) for w in F.sliding_windows(data_num)}
    return chain, per_window, verts[:400][None].expand(4, -1, -1).contiguous()
sgd = lambda chain: torch.optim.SGD(chain.parameters(), lr=2e-6)
def run_a():
    chain, wins, ov = problem()
    def window_views(index, vid, step): return wins[tuple(index)][vid]
    window_views.n_views = n_views
    torch.manual_seed(11)
    F.fit_sequence_video(renb, window_views, chain, 0.4, 1.5, data_num, '1234', outer_iters=outer, sub_iters=sub, obj_verts=ov, optimizer=sgd(chain))
    return chain
def run_b():
    chain, wins, ov = problem()
    opt = sgd(chain)
    torch.manual_seed(11)
    for it in range(outer):
        for index in F.sliding_windows(data_num):
            for s_ in range(sub):
                for vid in range(n_views):
                    later = it + s_ + vid > 0
                    F.fit_step(renb, wins[tuple(index)][vid], chain, opt, 0.4, 1.5, '1234', index=index,
                               smooth_ends=(later and index[0] == 0, later and index[-1] == data_num - 1), obj_verts_for_stable=ov)
    return chain
runs = {}
for name, fn in (('A1', run_a), ('B1', run_b), ('A2', run_a), ('B2', run_b)):
    runs[name] = fn()
    torch.cuda.synchronize()
moved = max(float((a.detach() - a.detach().round()).abs().max()) for a in runs['A1'].parameters())
dist = lambda x, y: max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(x.parameters(), y.parameters()))
names = list(runs)
for i in range(4):
    for k in range(i + 1, 4):
        print(names[i], names[k], '%.3e' % (dist(runs[names[i]], runs[names[k]]) / moved))

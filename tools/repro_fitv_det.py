"""Run-to-run reproducibility of a fitting_video window step: checksums of the loss terms and of every leaf's gradient bits."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, bench
from honerf_amd import fitting as F
dev = torch.device('cuda')
h = lambda x: int(x.detach().contiguous().view(torch.int32).to(torch.int64).sum().item()) & 0xffffffff
runs = []
names = ['obj_rot', 'obj_trans', 'palm_rot', 'palm_trans', 'joint', 'palm_angle']
for rep in range(int(os.environ.get('REPS', '6'))):
    ren, nets, chain, views, verts = bench.build_fit(dev, 41, 4, bench.VID_RAYS, 'f16x3', halo=True)
    with torch.no_grad():
        for i, p in enumerate(chain.parameters()):
            p.add_(1e-2 * torch.randn(p.shape, generator=torch.Generator().manual_seed(20 + i)).to(dev))
    opt = F.make_optimizer(chain, video=True)
    rows = []
    for k in range(6):
        tr = torch.rand(4 * bench.VID_RAYS, 1, generator=torch.Generator().manual_seed(300 + k)).to(dev)
        terms = F.fit_step(ren, views[k % len(views)], chain, opt, bench.NEAR, bench.FAR, '1234', index=[0, 1, 2, 3], smooth_ends=(k > 0, False),
                           obj_verts_for_stable=verts[:, :400], t_rand=tr)
        torch.cuda.synchronize()
        rows.append(({kk: h(v) for kk, v in terms.items()}, [h(p.grad) for p in chain.parameters()], [h(p) for p in chain.parameters()], h(ren.last_z_vals)))
    runs.append(rows)
for rep in range(1, len(runs)):
    for k in range(6):
        a, b = runs[0][k], runs[rep][k]
        if a != b:
            what = ['term_' + kk for kk in a[0] if a[0][kk] != b[0][kk]] + ['g_' + names[i] for i in range(6) if a[1][i] != b[1][i]]
            what += ['p_' + names[i] for i in range(6) if a[2][i] != b[2][i]] + (['z'] if a[3] != b[3] else [])
            print('rep %d first differs from rep 0 at step %d in: %s' % (rep, k, ' '.join(what)))
            break
    else:
        print('rep %d == rep 0 in every step (bits)' % rep)

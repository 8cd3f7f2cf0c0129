"""Run-to-run reproducibility of fitting_single steps: per step, a checksum of every leaf's gradient bits and of the final depths."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, bench
from honerf_amd import fitting as F
dev = torch.device('cuda')
h = lambda x: int(x.detach().contiguous().view(torch.int32).to(torch.int64).sum().item()) & 0xffffffff
pipelined = os.environ.get('PIPE', '1') == '1'
runs = []
for rep in range(int(os.environ.get('REPS', '4'))):
    ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True)
    with torch.no_grad():
        for i, p in enumerate(chain.parameters()):
            p.add_(4e-3 * torch.randn(p.shape, generator=torch.Generator().manual_seed(70 + i)).to(dev))
    opt = F.make_optimizer(chain, video=False)
    trs = [torch.rand(bench.FIT_RAYS, 1, generator=torch.Generator().manual_seed(190 + k)).to(dev) for k in range(12)]
    rows = []
    for k in range(12):
        terms = F.fit_step(ren, views[k % 8], chain, opt, bench.NEAR, bench.FAR, '12', t_rand=trs[k], pipelined=pipelined)
        F.finish_pipeline(opt); torch.cuda.synchronize()
        rows.append((h(ren.last_z_vals), h(terms['loss']), [h(p.grad) for p in chain.parameters()], [h(p) for p in chain.parameters()]))
    runs.append(rows)
names = ['obj_rot', 'obj_trans', 'palm_rot', 'palm_trans', 'joint', 'palm_angle']
for rep in range(1, len(runs)):
    for k in range(12):
        a, b = runs[0][k], runs[rep][k]
        if a != b:
            what = []
            if a[0] != b[0]: what.append('z')
            if a[1] != b[1]: what.append('loss')
            what += ['g_' + names[i] for i in range(6) if a[2][i] != b[2][i]]
            what += ['p_' + names[i] for i in range(6) if a[3][i] != b[3][i]]
            print('rep %d first differs from rep 0 at step %d in: %s' % (rep, k, ' '.join(what)))
            break
    else:
        print('rep %d == rep 0 in every step (bits)' % rep)

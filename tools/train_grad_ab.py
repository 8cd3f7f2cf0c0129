"""The parameter gradients of ONE training iteration (tools/train_step_bench.py's batch, no optimiser step) through the fused parameter-gradient
path (default) and the generic launch sequence (HN_TRAIN_FUSED=0), in two child processes: per parameter the largest |gradient| and the
relative difference.     python tools/train_grad_ab.py <obj|hand> [compact 0|1] [rays]"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))


def child(kind, compact, n_rays, out):
    import numpy as np
    import torch
    import train_step_bench as B
    from honerf_amd import training
    dev = torch.device('cuda:0')
    ren, synth = B.build(kind, dev)
    ren.precision = 'f16x3'
    ren.train_compact = bool(compact)
    ren.pack_eval_only = True
    o, d, ex = B.rays(kind, synth, n_rays, dev)
    g = torch.Generator(device='cpu').manual_seed(5)
    true_rgb = torch.rand(n_rays, 3, generator=g).to(dev)
    true_mask = (torch.rand(n_rays, 1, generator=g) > 0.3).float().to(dev)
    t_rand = torch.rand(n_rays, 1, generator=g).to(dev)
    ren.mark_parameters_changed()
    out_ = training.render_train(ren, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], None, ex['Ro'], ex['To'], t_rand=t_rand)
    terms = training.train_loss(out_, true_rgb, true_mask, 1.0, 1.0)
    terms['loss'].backward()
    torch.cuda.synchronize()
    names, grads = [], {}
    for mod, pre in ((ren.sdf_network, 'sdf'), (ren.color_network, 'color'), (ren.deviation_network, 'var')):
        for k, p in mod.named_parameters():
            if p.grad is not None:
                grads['%s.%s' % (pre, k)] = p.grad.detach().cpu().numpy()
    np.savez(out, loss=float(terms['loss']), **grads)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    import numpy as np
    kind = sys.argv[1] if len(sys.argv) > 1 else 'hand'
    compact = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 441
    outs = {}
    for flag in ('1', '0'):
        path = '/tmp/train_grad_ab_%s.npz' % flag
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', kind, str(compact), str(n_rays), path], env=dict(os.environ, HN_TRAIN_FUSED=flag))
        outs[flag] = np.load(path)
    a, b = outs['1'], outs['0']
    print('%s, compact %d, %d rays: loss fused %.6f generic %.6f' % (kind, compact, n_rays, a['loss'], b['loss']))
    for k in a.files:
        if k == 'loss':
            continue
        x, y = a[k], b[k]
        print('%-28s max|g| fused %.3e generic %.3e  rel diff %.3e  finite %s/%s' % (k, np.abs(x).max(), np.abs(y).max(), np.abs(x - y).max() / max(np.abs(y).max(), 1e-30),
                                                                                      np.isfinite(x).all(), np.isfinite(y).all()))

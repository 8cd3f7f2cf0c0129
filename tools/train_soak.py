"""The training loop over many iterations, fused parameter-gradient path (default) beside the generic launch sequence (HN_TRAIN_FUSED=0), in two
child processes on the same batches: loss every 25 iterations, finiteness of every parameter at the end, samples the hand adjoint dropped.
   python tools/train_soak.py <obj|hand> [iterations]"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))


def child(kind, iters, out):
    import json
    import torch
    import train_step_bench as B
    from honerf_amd import lib as L
    from honerf_amd import training
    dev = torch.device('cuda:0')
    ren, synth = B.build(kind, dev)
    ren.precision = 'f16x3'
    n_rays = 441
    opt = training.make_optimizer(ren, 5e-4)
    L.dropped_samples(reset=True)
    g = torch.Generator(device='cpu').manual_seed(11)
    losses = []
    for it in range(iters):
        o, d, ex = B.rays(kind, synth, n_rays, dev, seed=100 + it % 16)       # 16 batches in rotation
        gb = torch.Generator(device='cpu').manual_seed(200 + it % 16)
        true_rgb = torch.rand(n_rays, 3, generator=gb).to(dev)
        true_mask = (torch.rand(n_rays, 1, generator=gb) > 0.3).float().to(dev)
        t_rand = torch.rand(n_rays, 1, generator=g).to(dev)
        terms = training.train_step(ren, opt, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], ex['Ro'], ex['To'], true_rgb, true_mask, 1.0, 1.0, t_rand=t_rand)
        if it % 25 == 0 or it == iters - 1:
            losses.append((it, float(terms['loss'].detach())))
    finite = all(bool(torch.isfinite(p).all()) for p in training.trainable_parameters(ren))
    json.dump({'losses': losses, 'finite': finite, 'dropped': L.dropped_samples()}, open(out, 'w'))


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    import json
    kind = sys.argv[1] if len(sys.argv) > 1 else 'hand'
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    res = {}
    for flag in ('1', '0'):
        path = '/tmp/train_soak_%s.json' % flag
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', kind, str(iters), path], env=dict(os.environ, HN_TRAIN_FUSED=flag))
        res[flag] = json.load(open(path))
    print('%s nets, %d iterations of training.train_step (Adam 5e-4, 16 batches of 441 rays in rotation)' % (kind, iters))
    print('iteration   loss fused      loss generic')
    for (i, a), (_, b) in zip(res['1']['losses'], res['0']['losses']):
        print('%9d   %.6f      %.6f' % (i, a, b))
    print('all parameters finite: fused %s, generic %s; samples dropped by the hand adjoint: fused %d, generic %d (the generic sequence drops none: fp32)' % (
        res['1']['finite'], res['0']['finite'], res['1']['dropped'], res['0']['dropped']))

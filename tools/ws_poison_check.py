"""Uninitialised-read check: HONERF_POISON_WORKSPACES=1 fills every renderer workspace with NaNs before each use; a training
backward / fitting step whose results then contain a NaN read something the call did not write.
   HONERF_POISON_WORKSPACES=1 python tools/ws_poison_check.py"""
import os, sys
os.environ['HONERF_POISON_WORKSPACES'] = '1'
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from honerf_amd import training
from honerf_amd.nets import RenderingNetwork, RenderingNetwork_OBJ, SDFNetwork, SDFNetwork_OBJ, SingleVarianceNetwork
from honerf_amd.renderer import NeuSRenderer
dev = torch.device('cuda:0')
for kind in ('obj', 'hand'):
    g = np.load(os.path.join(R, 'tests', 'golden', 'train_%s.npz' % kind))
    c = lambda k: torch.tensor(g[k], dtype=torch.float32, device=dev)
    for compact in ((False,) if kind == 'obj' else (False, True)):
        if kind == 'obj':
            sdf_net, col_net, var = SDFNetwork_OBJ().to(dev), RenderingNetwork_OBJ().to(dev), SingleVarianceNetwork(0.3).to(dev)
        else:
            sdf_net, col_net, var = SDFNetwork().to(dev), RenderingNetwork(use_gradients=True).to(dev), SingleVarianceNetwork(0.3).to(dev)
        sdf_net.reset_parameters(21); col_net.reset_parameters(22)
        ren = NeuSRenderer(sdf_net, var, col_net, kind, int(g['n_samples']), int(g['n_importance']), 0, 4, 1.0)
        ren.train_compact = compact
        if kind == 'obj':
            args = (None, None, None, c('Ro').requires_grad_(True), c('To').requires_grad_(True))
        else:
            args = (c('bt_inv').requires_grad_(True), c('T_pose'), None, None, None)
        for rep in range(2):
            for p in training.trainable_parameters(ren):
                p.grad = None
            out = training.render_train(ren, c('rays_o'), c('rays_d'), float(g['near']), float(g['far']), *args, t_rand=c('t_rand'))
            terms = training.train_loss(out, c('true_rgb'), c('true_mask'), float(g['igr_weight']), float(g['mask_weight']))
            terms['loss'].backward()
            bad = [i for i, p in enumerate(training.trainable_parameters(ren)) if not torch.isfinite(p.grad).all()]
            print('%s compact=%s rep %d: loss %.6f, tensors with non-finite gradients: %s' % (kind, compact, rep, float(terms['loss']), bad))

#!/usr/bin/env python3
"""Generate tests/golden/train_{obj,hand}.npz: one iteration of exp_runner.train's inner loop (exp_runner.py:196-229)
run on the REFERENCE's own modules on CPU -- `NeuSRenderer.render`, the loss of :202-212 (colour L1, mask BCE, eikonal;
the VGG term is off, as before `vgg_start`), `loss.backward()` -- and the gradients autograd leaves on every parameter
of sdf_network / color_network / deviation_network.  Run in the build container only:

    python tests/golden/make_golden_train.py

Weights come from honerf_amd.synth (as in make_golden.py); a fixture is inputs + the reference's outputs.  To keep the
fixtures small the gradient of every `weight_v` is stored as every 8th column plus its full row sums and column sums;
`weight_g`, `bias` and `variance` gradients are stored whole.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)

COL_STEP = 8
IGR_WEIGHT, MASK_WEIGHT = 1.0, 1.0      # confs/wmask_realhand_hand1.conf:32-33


def train_iteration(ren, o, d, bt_inv, T_pose, Ro, To, true_rgb, true_mask, near, far, seed):
    torch.manual_seed(seed)
    t_rand = torch.rand([o.shape[0], 1])
    torch.manual_seed(seed)                      # the reference draws the same jitter inside render
    cap = mg.Capture(ren, 'render_core')
    out = ren.render(o, d, near, far, bt_inv, T_pose, None, Ro, To, 0)
    (a, _core), = cap.calls
    # exp_runner.py:202-212
    true_mask = (true_mask > 0.5).float()
    mask_sum = true_mask.sum() + 1e-5
    color_error = (out['color_fine'] - true_rgb) * true_mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / mask_sum
    eikonal_loss = out['gradient_error']
    mask_loss = F.binary_cross_entropy(out['weight_sum'].clip(1e-3, 1.0 - 1e-3), true_mask)
    loss = color_fine_loss + mask_loss * MASK_WEIGHT
    loss = loss + eikonal_loss * IGR_WEIGHT
    loss.backward()
    return t_rand, a[5], out, dict(loss=loss, color_fine_loss=color_fine_loss, mask_loss=mask_loss, eikonal_loss=eikonal_loss)


def grads_of(prefix, module):
    rec = {}
    for name, p in module.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        key = '%s.%s' % (prefix, name)
        if name.endswith('weight_v'):
            rec[key + '.cols'] = g[:, ::COL_STEP].contiguous()
            rec[key + '.rowsum'] = g.sum(dim=1)
            rec[key + '.colsum'] = g.sum(dim=0)
        else:
            rec[key] = g
    return rec


def main():
    torch.manual_seed(0)
    _emb, nets = mg.build_nets()
    g = torch.Generator().manual_seed(99)

    # object: the rays / pose of render_obj_64_64
    cam = mg.synth.front_camera(dist=1.0, focal=2.0)
    rng = np.random.RandomState(7)
    xy = (rng.rand(40, 2).astype(np.float32) - 0.5) * 1.2
    o, d = mg.rays_for(cam, xy)
    R_obj, t_obj = mg.synth.synth_obj_pose(2, center=(0.02, -0.01, 0.0))
    Ro = torch.from_numpy(R_obj).T.contiguous()
    To = torch.from_numpy(t_obj)
    true_rgb = torch.rand(40, 3, generator=g)
    true_mask = (torch.rand(40, 1, generator=g) > 0.3).float()
    ren = mg.rr.NeuSRenderer(nets['sdf_obj'], nets['var_obj'], nets['color_obj'], 'obj', 64, 64, 0, 4, 1.0)
    t_rand, z, out, terms = train_iteration(ren, o, d, torch.zeros(21, 4, 4), torch.zeros(21, 3), Ro, To, true_rgb, true_mask, 0.4, 1.5, 5)
    rec = {}
    rec.update(grads_of('sdf', nets['sdf_obj']))
    rec.update(grads_of('color', nets['color_obj']))
    rec.update(grads_of('var', nets['var_obj']))
    mg.save('train_obj', rays_o=o, rays_d=d, Ro=Ro, To=To, t_rand=t_rand, near=0.4, far=1.5, n_samples=64, n_importance=64,
            true_rgb=true_rgb, true_mask=true_mask, z_vals=z, igr_weight=IGR_WEIGHT, mask_weight=MASK_WEIGHT, col_step=COL_STEP,
            color_fine=out['color_fine'], weight_sum=out['weight_sum'], **terms, **rec)

    # hand: the rays / pose of render_hand_64_64
    o, d, bt_inv, T_pose, _joints = mg.hand_scene_rays(32, 9)
    true_rgb = torch.rand(32, 3, generator=g)
    true_mask = (torch.rand(32, 1, generator=g) > 0.3).float()
    ren = mg.rr.NeuSRenderer(nets['sdf_hand'], nets['var_hand'], nets['color_hand'], 'hand', 64, 64, 0, 4, 1.0)
    t_rand, z, out, terms = train_iteration(ren, o, d, bt_inv, T_pose, None, None, true_rgb, true_mask, 0.4, 1.5, 6)
    rec = {}
    rec.update(grads_of('sdf', nets['sdf_hand']))
    rec.update(grads_of('color', nets['color_hand']))
    rec.update(grads_of('var', nets['var_hand']))
    mg.save('train_hand', rays_o=o, rays_d=d, bt_inv=bt_inv, T_pose=T_pose, t_rand=t_rand, near=0.4, far=1.5, n_samples=64,
            n_importance=64, true_rgb=true_rgb, true_mask=true_mask, z_vals=z, igr_weight=IGR_WEIGHT, mask_weight=MASK_WEIGHT,
            col_step=COL_STEP, color_fine=out['color_fine'], weight_sum=out['weight_sum'], **terms, **rec)


    # the order in which the reference's modules enumerate their parameters (exp_runner.py:107-110 hands them to Adam in
    # this order; optimiser state dicts are index-based)
    order = {}
    for k in ('sdf_obj', 'color_obj', 'sdf_hand', 'color_hand'):
        order[k] = np.array(['%s %s' % (n, 'x'.join(str(d) for d in p.shape)) for n, p in nets[k].named_parameters()])
    np.savez_compressed(os.path.join(HERE, 'param_order.npz'), **order)


if __name__ == '__main__':
    main()

"""ORACLE (test infrastructure, not product): one iteration of exp_runner.train's inner loop (exp_runner.py:196-229)
restated over oracle.render.render_single -- render, the loss of :202-212 without the VGG term, backward by torch
autograd into the leaves `lin{l}.weight_g / weight_v / bias` of both networks and `variance`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  Parity status: PINNED --
tests/test_oracle_golden.py::test_train_iteration_golden checks it against tests/golden/train_{obj,hand}.npz, produced
by running the reference's own modules and autograd (tests/golden/make_golden_train.py).
"""
import torch
import torch.nn.functional as F

from . import render as orr
from .nets import Field, as_t


def trainable_field(kind, sdf_sd, color_sd, variance, scale=1.0, dtype=torch.float32):
    """An oracle Field whose folded matrices are differentiable functions of leaf tensors in the reference's
    state-dict layout (old-style weight norm, utils/fields.py:113-121): returns (field, {name: leaf})."""
    leaves = {}

    def build(sd, prefix):
        layers, l = [], 0
        while ('lin%d.bias' % l) in sd:
            g = as_t(sd['lin%d.weight_g' % l]).to(dtype).clone().requires_grad_(True)
            v = as_t(sd['lin%d.weight_v' % l]).to(dtype).clone().requires_grad_(True)
            b = as_t(sd['lin%d.bias' % l]).to(dtype).clone().requires_grad_(True)
            leaves['%s.lin%d.weight_g' % (prefix, l)] = g
            leaves['%s.lin%d.weight_v' % (prefix, l)] = v
            leaves['%s.lin%d.bias' % (prefix, l)] = b
            layers.append((torch._weight_norm(v, g, 0), b))
            l += 1
        return layers

    f = Field.__new__(Field)
    f.kind, f.scale = kind, scale
    f.sdf, f.color = build(sdf_sd, 'sdf'), build(color_sd, 'color')
    f.variance = torch.tensor(float(variance), dtype=dtype, requires_grad=True)
    leaves['var.variance'] = f.variance
    return f, leaves


def train_loss(out, true_rgb, true_mask, igr_weight, mask_weight):
    """exp_runner.py:202-212 (VGG term off)."""
    true_mask = (true_mask > 0.5).to(out['color_fine'].dtype)
    mask_sum = true_mask.sum() + 1e-5
    color_error = (out['color_fine'] - true_rgb) * true_mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / mask_sum
    mask_loss = F.binary_cross_entropy(out['weight_sum'].clip(1e-3, 1.0 - 1e-3), true_mask)
    eikonal_loss = out['gradient_error']
    loss = color_fine_loss + mask_loss * mask_weight + eikonal_loss * igr_weight
    return dict(loss=loss, color_fine_loss=color_fine_loss, mask_loss=mask_loss, eikonal_loss=eikonal_loss)


def train_iteration(field, leaves, rays_o, rays_d, near, far, t_rand, n_samples, n_importance, true_rgb, true_mask,
                    igr_weight=1.0, mask_weight=1.0, bt_inv=None, T_pose=None, Ro=None, To=None):
    """-> (render dict, loss terms, {leaf name: gradient})."""
    out = orr.render_single(field, rays_o, rays_d, near, far, t_rand, n_samples, n_importance, 4, bt_inv=bt_inv, T_pose=T_pose,
                            Ro=Ro, To=To)
    terms = train_loss(out, true_rgb, true_mask, igr_weight, mask_weight)
    names = list(leaves)
    grads = torch.autograd.grad(terms['loss'], [leaves[k] for k in names], allow_unused=True)
    return out, terms, {k: g for k, g in zip(names, grads)}


def core_iteration(field, leaves, rays_o, rays_d, z_vals, sample_dist, true_rgb, true_mask, igr_weight=1.0, mask_weight=1.0,
                   bt_inv=None, T_pose=None, Ro=None, To=None):
    """The same iteration on GIVEN depths: render_core (utils/renderer.py:107-177) + loss + backward, in the dtype of the
    field's leaves.  In float64 this is the 'exact' value both fp32 paths approximate (the noise-floor entries of
    tests/test_training.py).  -> (render dict, loss terms, {leaf name: gradient})."""
    dt = field.variance.dtype
    c = lambda x: None if x is None else as_t(x).to(dt)
    o, d, z = c(rays_o), c(rays_d), c(z_vals)
    if field.kind == 'obj':
        o, d = orr.obj_local(o, d, c(Ro), c(To))
    B, S = z.shape
    mid_z, dists = orr.mid_points(z, sample_dist)
    pts = orr._pts(o, d, mid_z).reshape(-1, 3)
    dirs = d[:, None, :].expand(B, S, 3).reshape(-1, 3)
    sdf, grad, rgb = field.evaluate(pts, dirs, c(bt_inv), c(T_pose))
    inv_s = torch.exp(field.variance * 10.0).clip(1e-6, 1e6)      # utils/fields.py:248-249, utils/renderer.py:144
    alpha, cc = orr.sdf_to_alpha(sdf, grad, dirs, dists.reshape(-1, 1), inv_s)
    alpha, cc = alpha.reshape(B, S), cc.reshape(B, S)
    w, color = orr.composite_single(alpha, cc, rgb.reshape(B, S, 3))
    out = {'color_fine': color, 'weight_sum': w.sum(dim=-1, keepdim=True), 'gradient_error': orr.eikonal(grad, (B, S))}
    terms = train_loss(out, c(true_rgb), c(true_mask), igr_weight, mask_weight)
    names = list(leaves)
    grads = torch.autograd.grad(terms['loss'], [leaves[k] for k in names], allow_unused=True)
    return out, terms, {k: g for k, g in zip(names, grads)}

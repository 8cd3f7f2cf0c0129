"""ORACLE (test infrastructure, not product): CPU restatement of the HO-NeRF
field networks in plain PyTorch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  Parity status: PINNED -- tests/test_oracle_golden.py
checks every function here against vectors produced by importing the reference
itself (tests/golden/make_golden.py, run in the build container).

Each function cites the reference lines it restates (paths relative to the
reference checkout).  Networks are held as plain lists of (W, b) with the
weight-norm already folded; `mlp_from_state_dict` reads the reference's
state-dict layout.
"""
import math

import torch
import torch.nn.functional as F

HAND_CUTOFF = (0.08, 0.03, 0.03, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02,
               0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02)
TAU = 200.0
SKIP = 4


def as_t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float32)


def mlp_from_state_dict(sd, requires_grad=False):
    """[(W, b)] from `lin{l}.weight_g / weight_v / bias` (old-style
    nn.utils.weight_norm, dim=0: utils/fields.py:120-121): W = g * v / ||v||_row."""
    layers = []
    l = 0
    while ('lin%d.bias' % l) in sd:
        b = as_t(sd['lin%d.bias' % l]).float().clone()
        if ('lin%d.weight_g' % l) in sd:
            g = as_t(sd['lin%d.weight_g' % l]).float()
            v = as_t(sd['lin%d.weight_v' % l]).float()
            W = torch._weight_norm(v, g, 0)      # what nn.utils.weight_norm evaluates: v * (g / ||v||)
        else:
            W = as_t(sd['lin%d.weight' % l]).float().clone()
        W = W.detach().clone().requires_grad_(requires_grad)
        b = b.detach().clone().requires_grad_(requires_grad)
        layers.append((W, b))
        l += 1
    return layers


def embed(x, L):
    """utils/fields.py:13-20.  [...,C] -> [...,2CL]; per channel [sin(2^0 x)..
    sin(2^(L-1) x), cos(2^0 x)..cos(2^(L-1) x)]; no pi factor."""
    freq = 2.0 ** torch.arange(L, dtype=torch.float32)
    ang = x[..., None] * freq                       # [...,C,L]
    enc = torch.stack([ang.sin(), ang.cos()], dim=-2)   # [...,C,2,L]
    return enc.reshape(*x.shape[:-1], -1)


def softplus100(x):
    """nn.Softplus(beta=100), threshold 20 (utils/fields.py:125, 310)."""
    return F.softplus(x, beta=100.0, threshold=20.0)


def bone_coords(pts, bt_inv, T_pose):
    """utils/fields.py:22-36 (and the batched copy :38-52 via broadcasting).

    pts [...,N,3]; bt_inv [...,21,4,4]; T_pose [...,21,3]  (leading dims = frames)
    -> v [...,N,21,1], r [...,N,21,3], h [...,N,21,1]
    """
    cutoff = torch.tensor(HAND_CUTOFF, dtype=torch.float32).reshape(21, 1)
    R = bt_inv[..., None, :, :3, :3]                 # [...,1,21,3,3]
    t = bt_inv[..., None, :, :3, 3]                  # [...,1,21,3]
    q = torch.matmul(R, pts[..., :, None, :, None])[..., 0] + t
    q = q - T_pose[..., None, :, :]
    v = torch.norm(q, dim=-1, p=2).unsqueeze(-1)
    r = q / v                                        # no epsilon: NaN on a joint (SURVEY B-10)
    h = 1.0 - torch.sigmoid(TAU * (v - cutoff))
    return v, r, h


def hand_features(pts, bt_inv, T_pose, v_freqs=10, r_freqs=7):
    """utils/fields.py:134-147: per bone [v, enc(v), r, enc(r)] * h, bone-major
    flatten -> [M, 21*66].  pts may be [N,3] with bt_inv [21,4,4] or [F,N,3]
    with bt_inv [F,21,4,4] (use_batch path)."""
    v, r, h = bone_coords(pts, bt_inv, T_pose)
    v = v.reshape(-1, 21, 1)
    r = r.reshape(-1, 21, 3)
    h = h.reshape(-1, 21, 1)
    per_bone = torch.cat([v, embed(v, v_freqs), r, embed(r, r_freqs)], dim=-1) * h
    return per_bone.flatten(start_dim=-2), r, h


def hand_sdf_forward(mlp, pts, bt_inv, T_pose):
    """SDFNetwork.forward, utils/fields.py:132-156 -> (out[M,257], feat[M,1386])."""
    feat, _, _ = hand_features(pts, bt_inv, T_pose)
    x = feat
    n = len(mlp)
    for l, (W, b) in enumerate(mlp):
        if l == SKIP:
            x = torch.cat([x, feat], dim=1) / math.sqrt(2.0)
        x = F.linear(x, W, b)
        if l < n - 1:
            x = softplus100(x)
    return x, feat


def obj_sdf_forward(mlp, pts, scale=1.0, v_freqs=10):
    """SDFNetwork_OBJ.forward, utils/fields.py:316-328 -> out[M,257]."""
    inp = torch.cat([pts, embed(pts, v_freqs)], dim=-1)
    x = inp
    n = len(mlp)
    for l, (W, b) in enumerate(mlp):
        if l == SKIP:
            x = torch.cat([x, inp], dim=1) / math.sqrt(2.0)
        x = F.linear(x, W, b)
        if l < n - 1:
            x = softplus100(x)
    return torch.cat([x[:, :1] / scale, x[:, 1:]], dim=-1)


def sdf_and_gradient(fn, pts, create_graph=True):
    """`.gradient()` of both SDF nets (utils/fields.py:165-177, 336-347):
    d sdf / d pts by autograd with create_graph=True.  fn(pts) -> full output;
    returns (output, gradient [M,3]).  Unlike the reference (SURVEY B-4) the
    forward is evaluated once; the values are identical."""
    pts = pts if pts.requires_grad else pts.detach().requires_grad_(True)
    out = fn(pts)
    extra = None
    if isinstance(out, tuple):
        out, extra = out
    y = out[..., :1]
    g = torch.autograd.grad(y, pts, torch.ones_like(y), create_graph=create_graph,
                            retain_graph=True, only_inputs=True)[0]
    return out, extra, g


def _color_tail(mlp, x):
    n = len(mlp)
    for l, (W, b) in enumerate(mlp):
        x = F.linear(x, W, b)
        if l < n - 1:
            x = F.relu(x)
    return torch.sigmoid(x)


def hand_color(mlp, feat, feature_vec, grads, g_freqs=4):
    """RenderingNetwork.forward, utils/fields.py:222-240 (d and h unused)."""
    gin = torch.cat([grads, embed(grads, g_freqs)], dim=-1)
    return _color_tail(mlp, torch.cat([feat, feature_vec, gin], dim=-1))


def obj_color(mlp, pts, dirs, feature_vec, grads, v_freqs=10, r_freqs=4, g_freqs=4):
    """RenderingNetwork_OBJ.forward, utils/fields.py:387-405."""
    pin = torch.cat([pts, embed(pts, v_freqs)], dim=-1)
    din = torch.cat([dirs, embed(dirs, r_freqs)], dim=-1)
    gin = torch.cat([grads, embed(grads, g_freqs)], dim=-1)
    return _color_tail(mlp, torch.cat([pin, din, feature_vec, gin], dim=-1))


def inv_s_from_variance(variance):
    """SingleVarianceNetwork.forward + clip (utils/fields.py:248-249,
    utils/renderer.py:144): clip(exp(10*variance), 1e-6, 1e6), a scalar."""
    return torch.exp(as_t(variance).float() * 10.0).clip(1e-6, 1e6)


class Field:
    """One (sdf, colour, variance) triple; kind 'obj' or 'hand'."""

    def __init__(self, kind, sdf_sd, color_sd, variance=0.3, scale=1.0, requires_grad=False):
        self.kind = kind
        self.sdf = mlp_from_state_dict(sdf_sd, requires_grad)
        self.color = mlp_from_state_dict(color_sd, requires_grad)
        self.variance = as_t(variance).float()
        self.scale = scale

    def inv_s(self):
        return inv_s_from_variance(self.variance)

    def sdf_only(self, pts, bt_inv=None, T_pose=None):
        """.sdf(): utils/fields.py:158-160, 330-331 -> [M,1]."""
        if self.kind == 'obj':
            return obj_sdf_forward(self.sdf, pts, self.scale)[:, :1]
        return hand_sdf_forward(self.sdf, pts, bt_inv, T_pose)[0][:, :1]

    def evaluate(self, pts, dirs, bt_inv=None, T_pose=None):
        """sdf, d sdf/d pts and colour at pts, as render_core /
        get_alpha_sample_color call the three modules (utils/renderer.py:
        130-142, 380-396; utils/renderer_batch.py:133-149).
        pts [M,3] (hand batched: [F,N,3] with bt_inv [F,21,4,4]); dirs [M,3].
        -> sdf [M,1], grad [M,3], rgb [M,3]"""
        if self.kind == 'obj':
            out, _, g = sdf_and_gradient(lambda p: obj_sdf_forward(self.sdf, p, self.scale), pts)
            rgb = obj_color(self.color, pts, dirs, out[:, 1:], g)
            return out[:, :1], g, rgb
        out, feat, g = sdf_and_gradient(lambda p: hand_sdf_forward(self.sdf, p, bt_inv, T_pose), pts)
        g = g.reshape(-1, 3)
        rgb = hand_color(self.color, feat, out[:, 1:], g)
        return out[:, :1], g, rgb

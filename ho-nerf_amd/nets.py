"""Parameter containers with the reference's constructor signatures and
state-dict layout, and the packed device-side field they feed.

The classes mirror utils/fields.py (Embedding :8-20, SDFNetwork :56-177,
RenderingNetwork :179-240, SingleVarianceNetwork :243-249, SDFNetwork_OBJ
:251-347, RenderingNetwork_OBJ :349-405): same constructor arguments, same
parameter names (``lin{l}.weight_g``, ``lin{l}.weight_v``, ``lin{l}.bias``,
``se3_refine``, ``variance``), so a reference checkpoint loads with
``load_state_dict`` unchanged.  They hold parameters only: every evaluation
goes through the HIP library (PackedField); there is no PyTorch forward.
"""
import ctypes
import math

import numpy as np
import weakref

import torch
import torch.nn as nn

from . import lib as _lib
from . import synth


class Embedding(nn.Module):
    """Placeholder for the reference's `barf_encoding` argument (utils/fields.py:8-20).
    The encoding itself is evaluated inside the field kernels."""

    def __init__(self):
        super().__init__()


class _WNLinear(nn.Module):
    """Parameters of one weight-normalised nn.Linear (old-style nn.utils.weight_norm)."""

    def __init__(self, cin, cout):
        super().__init__()
        # registration order of a weight-normalised nn.Linear (bias, weight_g, weight_v): `parameters()` then enumerates
        # like the reference's modules, so an optimiser state dict (index-based) is interchangeable with exp_runner's
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_g = nn.Parameter(torch.ones(cout, 1))
        self.weight_v = nn.Parameter(torch.zeros(cout, cin))


class _MLPParams(nn.Module):
    kind = None

    def _build(self, d_hidden, seed_kind):
        shapes = synth.layer_shapes(self.kind, d_hidden)
        self.num_layers = len(shapes) + 1
        for l, (out, cin) in enumerate(shapes):
            setattr(self, 'lin%d' % l, _WNLinear(cin, out))
        self.reset_parameters(0)

    def reset_parameters(self, seed=0):
        """Random init with the statistics of the reference constructors (geometric
        init for the SDF nets), drawn from honerf_amd.synth so that it is reproducible."""
        sd = synth.synth_state_dict(self.kind, seed)
        with torch.no_grad():
            for k, v in sd.items():
                mod, name = k.split('.')
                getattr(getattr(self, mod), name).copy_(torch.from_numpy(v))

    def layers(self):
        return [getattr(self, 'lin%d' % l) for l in range(self.num_layers - 1)]


def _filler_state_dict(kind):
    """A finite placeholder network of the conf shape (weight_v[:, 0] = 1, everything else 0) for the half of a
    PackedField that a stand-alone module call does not use."""
    sd = {}
    for l, (out, cin) in enumerate(synth.layer_shapes(kind, 256)):
        v = torch.zeros(out, cin)
        v[:, 0] = 1.0
        sd['lin%d.weight_v' % l] = v
        sd['lin%d.weight_g' % l] = torch.zeros(out, 1)
        sd['lin%d.bias' % l] = torch.zeros(out)
    return sd


class _FieldCallFn(torch.autograd.Function):
    """A stand-alone SDF-module call WITH a graph: (pts [n,3], bt_inv [F,21,4,4] | None, T_pose [F,21,3] | None, *the module's
    parameters) -> (sdf [n,1], d sdf / d pts [n,3]).  The reference's `.sdf()` is an nn.Module forward and its `.gradient()` an
    `autograd.grad(..., create_graph=True)` (utils/fields.py:158-177, 330-347): both can be differentiated again, w.r.t. the points,
    the hand's pose inputs and every parameter -- that is how the eikonal term reaches the weights.  Backward here = hn_field_param_bwd
    (the adjoint of one field evaluation incl. the second-order path through the gradient output, with d / d folded weights) +
    hn_weight_norm_bwd (the weight-norm chain rule), as the training iteration has them."""

    @staticmethod
    def forward(ctx, module, pts, bt_inv, T_pose, *params):
        pf = module._packed()
        p = _lib.f32(pts).reshape(-1, 3)
        bt = None if bt_inv is None else _lib.f32(bt_inv, p.device).reshape(-1, 21, 4, 4)
        tp = None if T_pose is None else _lib.f32(T_pose, p.device).reshape(-1, 21, 3)
        sdf, grad, _ = pf.evaluate(p, torch.zeros_like(p), 1, bt, tp)
        ctx.module, ctx.pf = module, pf
        ctx.shapes = (pts.shape, None if bt_inv is None else bt_inv.shape, None if T_pose is None else T_pose.shape)
        ctx.save_for_backward(p, *([bt, tp] if bt is not None else []))
        return sdf, grad

    @staticmethod
    def backward(ctx, g_sdf, g_grad):
        L = _lib
        module, pf = ctx.module, ctx.pf
        lib = pf.lib
        sv = ctx.saved_tensors
        p = sv[0]
        hand = len(sv) > 1
        bt, tp = (sv[1], sv[2]) if hand else (None, None)
        n, dev = p.shape[0], p.device
        nf = bt.shape[0] if hand else 1
        if hand and tp.shape[0] != nf:
            tp = tp.expand(nf, 21, 3).contiguous()
        gs = None if g_sdf is None else L.f32(g_sdf).reshape(n)
        gg = None if g_grad is None else L.f32(g_grad).reshape(n, 3)
        if gs is None and gg is None:
            return (None,) * (4 + len(module._call_params()))
        zeros3 = torch.zeros(n, 3, device=dev)
        # (the adjoint takes all three upstream gradients or the sdf's alone: a missing one is zero)
        gs = torch.zeros(n, device=dev) if gs is None else gs
        gg = zeros3 if gg is None else gg
        g_params = torch.zeros(lib.hn_field_param_floats(pf.handle), device=dev)
        g_pts, g_dir = torch.empty(n, 3, device=dev), torch.zeros(n, 3, device=dev)
        g_bt = torch.zeros(nf, 21, 4, 4, device=dev) if hand else None
        g_tp = torch.zeros(nf, 21, 3, device=dev) if hand else None
        need = lib.hn_field_bwd_workspace_bytes(pf.handle, n)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        L.check(lib.hn_field_param_bwd(pf.handle, L.ptr(p), L.ptr(zeros3), n, 1, L.ptr(bt), L.ptr(tp), nf, max(n // nf, 1), L.ptr(gs), L.ptr(gg),
                                       L.ptr(zeros3), L.ptr(g_params), L.ptr(g_pts), L.ptr(g_dir), L.ptr(g_bt), L.ptr(g_tp), L.ptr(ws), need,
                                       L.stream_ptr()), 'hn_field_param_bwd')
        # weight-norm chain rule (hn_weight_norm_bwd) into this module's (weight_g, weight_v, bias); the placeholder network of the
        # other half of the field receives its (zero) gradients into scratch
        keep = []
        mine, other = module.state_dict(), module._filler
        d_sdf, d_col = (_mlp_desc(mine, keep), _mlp_desc(other, keep)) if module._is_sdf else (_mlp_desc(other, keep), _mlp_desc(mine, keep))
        grads, outs = [], []
        for sd_, is_mine in ((mine, module._is_sdf), (other, not module._is_sdf)) if module._is_sdf else ((other, False), (mine, True)):
            d = L.MlpDesc()
            l = 0
            while ('lin%d.bias' % l) in sd_:
                v, b = sd_['lin%d.weight_v' % l], sd_['lin%d.bias' % l]
                dg, dv, db = torch.empty(v.shape[0], 1, device=dev), torch.empty(v.shape, device=dev), torch.empty(b.shape, device=dev)
                d.weight_g[l], d.weight_v[l], d.bias[l] = dg.data_ptr(), dv.data_ptr(), db.data_ptr()
                d.out_dim[l], d.in_dim[l] = v.shape
                keep += [dg, dv, db]
                if is_mine:
                    grads += [dg, dv, db]
                l += 1
            d.n_layers = l
            outs.append(d)
        L.check(lib.hn_weight_norm_bwd(pf.handle, ctypes.byref(d_sdf), ctypes.byref(d_col), L.ptr(g_params), ctypes.byref(outs[0]),
                                       ctypes.byref(outs[1]), L.stream_ptr()), 'hn_weight_norm_bwd')
        s_p, s_b, s_t = ctx.shapes
        g_tp_out = None
        if hand:
            g_tp_out = g_tp.reshape(s_t) if g_tp.numel() == int(torch.Size(s_t).numel()) else g_tp.sum(0).reshape(s_t)
        return (None, g_pts.reshape(s_p), g_bt.reshape(s_b) if hand else None, g_tp_out, *grads)


class _Standalone:
    """A module called on its own (utils/fields.py `forward` / `.sdf` / `.gradient`), outside a renderer: it packs
    itself (lazily, re-packed when its parameters change) next to a placeholder for the other network of the field.
    `.sdf()` / `.gradient()` of the SDF modules build a graph when grad mode is on and a point, pose input or parameter requires
    grad (`_FieldCallFn`); `forward()`'s feature columns and the colour modules' calls are forward only -- their differentiable path
    is the renderers' (autograd.DualRenderFn, training.SingleRenderFn)."""
    _field_kind = None      # 'obj' | 'hand'
    _is_sdf = True

    def _call_params(self):
        """(weight_g, weight_v, bias) per layer, the order _FieldCallFn.backward returns their gradients in."""
        ps = []
        for lin in self.layers():
            ps += [lin.weight_g, lin.weight_v, lin.bias]
        return ps

    def _wants_graph(self, *inputs):
        if not torch.is_grad_enabled():
            return False
        return any(isinstance(x, torch.Tensor) and x.requires_grad for x in inputs) or any(p.requires_grad for p in self._call_params())

    def _sdf_and_gradient(self, pts, bt_inv=None, T_pose=None):
        """(sdf [n,1], gradient [n,3]) with a graph."""
        return _FieldCallFn.apply(self, pts, bt_inv, T_pose, *self._call_params())

    def _packed(self):
        ver = params_version(self)
        if getattr(self, '_pf', None) is None or self._pf_ver != ver:
            other = {k: v.cuda() for k, v in _filler_state_dict(('color_' if self._is_sdf else 'sdf_') + self._field_kind).items()}
            object.__setattr__(self, '_filler', other)
            mine = self
            sdf, col = (mine, other) if self._is_sdf else (other, mine)
            object.__setattr__(self, '_pf', PackedField(self._field_kind, sdf, col, 0.3, scale=float(getattr(self, 'scale', 1.0))))
            object.__setattr__(self, '_pf_ver', ver)
        return self._pf


class SDFNetwork_OBJ(_MLPParams, _Standalone):
    """utils/fields.py:251-347."""
    kind = 'sdf_obj'
    _field_kind, _is_sdf = 'obj', True

    def forward(self, inputs):
        """utils/fields.py:316-328: [N,3] -> [N,257] = cat[sdf / scale, feature vector]."""
        pts = _lib.f32(inputs).reshape(-1, 3)
        sdf, _, _, feat = self._packed().evaluate(pts, torch.zeros_like(pts), 1, want_feat=True)
        return torch.cat([sdf, feat], dim=-1)

    def sdf(self, x):
        """utils/fields.py:330-331 -> [N,1]."""
        if self._wants_graph(x):
            return self._sdf_and_gradient(x.reshape(-1, 3))[0]
        return self._packed().sdf(x)

    def gradient(self, x):
        """utils/fields.py:336-347: d sdf / d x, [N,1,3] (analytic reverse sweep instead of autograd.grad); differentiable again, as
        the reference's create_graph=True result is (hn_field_param_bwd)."""
        if self._wants_graph(x):
            return self._sdf_and_gradient(x.reshape(-1, 3))[1].unsqueeze(1)
        pts = _lib.f32(x).reshape(-1, 3)
        _, grad, _ = self._packed().evaluate(pts, torch.zeros_like(pts), 1)
        return grad.unsqueeze(1)

    def __init__(self, barf_encoding=None, traindata_num=1, data_type='real', d_in=3, d_out=257, d_hidden=256,
                 n_layers=8, skip_in=(4,), v_multires=10, r_multires=4, bias=0.5, scale=1, geometric_init=True,
                 weight_norm=True, inside_outside=False):
        super().__init__()
        _require(d_in == 3 and d_out == 257 and d_hidden == 256 and n_layers == 8 and tuple(skip_in) == (4,)
                 and v_multires == 10 and weight_norm, 'SDFNetwork_OBJ: only the published conf shape is supported')
        self.scale = scale
        self._build(d_hidden, 'sdf_obj')
        se3 = torch.zeros((traindata_num, 6 + 3))
        se3[:, 0] = 1
        se3[:, 3] = 1
        self.se3_refine = nn.Parameter(se3)


class SDFNetwork(_MLPParams, _Standalone):
    """utils/fields.py:56-177 (hand)."""
    kind = 'sdf_hand'
    _field_kind, _is_sdf = 'hand', True

    def _frames(self, x, bt_inv, T_pose_21):
        pts = _lib.f32(x)
        bt = _lib.f32(bt_inv, pts.device).reshape(-1, 21, 4, 4)
        tp = _lib.f32(T_pose_21, pts.device).reshape(-1, 21, 3)
        if not self.use_batch:
            bt, tp = bt[:1], tp[:1]
        return pts.reshape(-1, 3), bt.contiguous(), tp.contiguous()

    def forward(self, x, bt_inv, T_pose_21):
        """utils/fields.py:132-156: x [N,3] (use_batch: [F,N,3] with bt_inv [F,21,4,4]) ->
        (out [M,257], xyz_feature [M,1386], r [M,21,3], h [M,21,1])."""
        pts, bt, tp = self._frames(x, bt_inv, T_pose_21)
        pf = self._packed()
        n = pts.shape[0]
        sdf, _, _, feat = pf.evaluate(pts, torch.zeros_like(pts), 1, bt, tp, want_feat=True)
        nf = bt.shape[0]
        if tp.shape[0] != nf:
            tp = tp.expand(nf, 21, 3).contiguous()
        X = torch.empty(n, 21 * 66, device=pts.device)
        r = torch.empty(n, 21, 3, device=pts.device)
        h = torch.empty(n, 21, 1, device=pts.device)
        _lib.check(pf.lib.hn_hand_features(_lib.ptr(pts), n, _lib.ptr(bt), _lib.ptr(tp), nf, max(n // nf, 1), _lib.ptr(X), _lib.ptr(r),
                                           _lib.ptr(h), _lib.stream_ptr()), 'hn_hand_features')
        return torch.cat([sdf, feat], dim=-1), X, r, h

    def _graph_inputs(self, x, bt_inv, T_pose_21):
        """The call's inputs as autograd sees them (no detach): points [M,3], bt_inv [F,21,4,4], T_pose [F,21,3]."""
        t = lambda v: v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=torch.float32, device=x.device if isinstance(x, torch.Tensor) else 'cuda')
        bt, tp = t(bt_inv).reshape(-1, 21, 4, 4), t(T_pose_21).reshape(-1, 21, 3)
        if not self.use_batch:
            bt, tp = bt[:1], tp[:1]
        return t(x).reshape(-1, 3), bt, tp

    def sdf(self, x, bt_inv, T_pose_21):
        """utils/fields.py:158-160 -> [M,1]."""
        if self._wants_graph(x, bt_inv, T_pose_21):
            return self._sdf_and_gradient(*self._graph_inputs(x, bt_inv, T_pose_21))[0]
        pts, bt, tp = self._frames(x, bt_inv, T_pose_21)
        return self._packed().sdf(pts, bt, tp)

    def gradient(self, x, bt_inv, T_pose_21):
        """utils/fields.py:165-177: d sdf / d x, [M,1,3] (use_batch: [F,1,N,3] squeezes to the same rows); differentiable again, as
        the reference's create_graph=True result is."""
        if self._wants_graph(x, bt_inv, T_pose_21):
            return self._sdf_and_gradient(*self._graph_inputs(x, bt_inv, T_pose_21))[1].unsqueeze(1)
        pts, bt, tp = self._frames(x, bt_inv, T_pose_21)
        _, grad, _ = self._packed().evaluate(pts, torch.zeros_like(pts), 1, bt, tp)
        return grad.unsqueeze(1)

    def __init__(self, barf_encoding=None, traindata_num=1, data_type='real', d_in=3, d_out=257, d_hidden=256,
                 n_layers=8, skip_in=(4,), v_multires=10, r_multires=7, bias=0.5, scale=1, geometric_init=True,
                 weight_norm=True, inside_outside=False, use_batch=False):
        super().__init__()
        _require(d_in == 3 and d_out == 257 and d_hidden == 256 and n_layers == 8 and tuple(skip_in) == (4,)
                 and v_multires == 10 and r_multires == 7 and weight_norm,
                 'SDFNetwork: only the published conf shape is supported')
        self.scale = scale
        self.use_batch = use_batch
        self._build(d_hidden, 'sdf_hand')
        se3 = torch.zeros((traindata_num, 6 + 3 + 20 + 7))
        se3[:, 0] = 1
        se3[:, 3] = 1
        self.se3_refine = nn.Parameter(se3)


def _color_forward(pf, x, view_dirs, feature_vectors, normals):
    x = _lib.f32(x)
    x = x.reshape(-1, x.shape[-1])
    n = x.shape[0]
    dev = x.device
    fv = _lib.f32(feature_vectors, dev).reshape(n, 256)
    nr = _lib.f32(normals, dev).reshape(n, 3)
    vd = None if view_dirs is None else _lib.f32(view_dirs, dev).reshape(n, 3)
    rgb = torch.empty(n, 3, device=dev)
    need = pf.lib.hn_color_forward_workspace_bytes(pf.handle, n)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    _lib.check(pf.lib.hn_color_forward(pf.handle, _lib.ptr(x), _lib.ptr(vd), _lib.ptr(fv), _lib.ptr(nr), n, _lib.ptr(rgb), _lib.ptr(ws),
                                       need, _lib.stream_ptr()), 'hn_color_forward')
    return rgb


class RenderingNetwork_OBJ(_MLPParams, _Standalone):
    """utils/fields.py:349-405."""
    kind = 'color_obj'
    _field_kind, _is_sdf = 'obj', False

    def forward(self, points, view_dirs, feature_vectors, normals, index=0):
        """utils/fields.py:387-405 -> rgb [N,3]."""
        return _color_forward(self._packed(), points, view_dirs, feature_vectors, normals)

    def __init__(self, barf_encoding=None, data_type='real', d_feature=256, d_in=3, d_out=3, d_hidden=256, n_layers=4,
                 weight_norm=True, v_multires=10, r_multires=4, grad_multires=4, squeeze_out=True,
                 use_gradients=False):
        super().__init__()
        _require(d_feature == 256 and d_in == 3 and d_out == 3 and d_hidden == 256 and n_layers == 4 and weight_norm
                 and v_multires == 10 and r_multires == 4 and grad_multires == 4 and squeeze_out,
                 'RenderingNetwork_OBJ: only the published conf shape is supported')
        self._build(d_hidden, 'color_obj')


class RenderingNetwork(_MLPParams, _Standalone):
    """utils/fields.py:179-240 (hand)."""
    kind = 'color_hand'
    _field_kind, _is_sdf = 'hand', False

    def forward(self, view_dirs, xyz_feature, feature_vectors, h, normals, index=0):
        """utils/fields.py:222-240 (view_dirs and h are accepted and unused, as there) -> rgb [N,3]."""
        return _color_forward(self._packed(), xyz_feature, None, feature_vectors, normals)

    def __init__(self, barf_encoding=None, data_type='real', d_feature=256, d_in=3, d_out=3, d_hidden=256, n_layers=4,
                 weight_norm=True, v_multires=10, r_multires=7, grad_multires=4, squeeze_out=True,
                 use_gradients=False):
        super().__init__()
        _require(d_feature == 256 and d_in == 3 and d_out == 3 and d_hidden == 256 and n_layers == 4 and weight_norm
                 and v_multires == 10 and r_multires == 7 and grad_multires == 4 and squeeze_out and use_gradients,
                 'RenderingNetwork: only the published conf shape (use_gradients=True) is supported')
        self._build(d_hidden, 'color_hand')


class SingleVarianceNetwork(nn.Module):
    """utils/fields.py:243-249."""

    def __init__(self, init_val):
        super().__init__()
        self.register_parameter('variance', nn.Parameter(torch.tensor(float(init_val))))

    def forward(self, x):
        """utils/fields.py:248-249: ones([len(x), 1]) * exp(10 variance) (a scalar broadcast: plain torch)."""
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)


def _require(cond, msg):
    if not cond:
        raise ValueError(msg)


def _state_of(module_or_sd):
    if isinstance(module_or_sd, dict):
        return module_or_sd
    return module_or_sd.state_dict()


def _mlp_desc(sd, keep):
    """hn_mlp_desc from a state dict (tensors or numpy); `keep` collects the device
    tensors so they outlive the call."""
    d = _lib.MlpDesc()
    l = 0
    while ('lin%d.bias' % l) in sd:
        b = _lib.f32(sd['lin%d.bias' % l])
        if ('lin%d.weight_v' % l) in sd:
            v = _lib.f32(sd['lin%d.weight_v' % l])
            g = _lib.f32(sd['lin%d.weight_g' % l])
            d.weight_g[l] = g.data_ptr()
            keep.append(g)
        else:                                   # a plain nn.Linear checkpoint
            v = _lib.f32(sd['lin%d.weight' % l])
            d.weight_g[l] = None
        keep += [v, b]
        d.weight_v[l] = v.data_ptr()
        d.bias[l] = b.data_ptr()
        d.out_dim[l], d.in_dim[l] = v.shape
        l += 1
    d.n_layers = l
    return d


class PackedField:
    """Device-resident, immutable packed weights of one field (hn_field).

    kind 'obj' | 'hand'; `sdf`, `color` are modules (ours or the reference's) or
    state dicts in the reference layout; `variance` a SingleVarianceNetwork, a
    tensor or a float."""

    def __init__(self, kind, sdf, color, variance, scale=None, precision=None, eval_only=False, device_variance=False):
        """`device_variance` (single-field renderers: `NeuSRenderer.field()`): a variance given as a device tensor stays there --
        `inv_s = clip(exp(10 variance), 1e-6, 1e6)` is formed on the current stream (hn_variance_to_inv_s) and handed to the library as
        a device scalar (hn_field_set_inv_s_device), so that packing a field does not wait for the device (a training iteration
        re-packs after every optimiser step: exp_runner.py:107-110 trains the variance too).  `.inv_s` / `.variance` then read the
        value back on first use; `.inv_s_t` is the device scalar.  Such a field serves the single-field renders only."""
        self.lib = _lib.load()
        self.kind = kind
        precision = precision or _lib.DEFAULT_PRECISION
        self.precision = precision
        self.eval_only = bool(eval_only)
        sdf_sd, col_sd = _state_of(sdf), _state_of(color)
        if isinstance(variance, nn.Module):
            variance = variance.variance
        self.inv_s_t = None
        self._variance_t = None
        if device_variance and isinstance(variance, torch.Tensor) and variance.is_cuda:
            v = variance.detach().reshape(1).float()
            self._variance_t = v
            self.inv_s_t = torch.empty(1, device=v.device, dtype=torch.float32)
            _lib.check(self.lib.hn_variance_to_inv_s(_lib.ptr(v.contiguous()), _lib.ptr(self.inv_s_t), _lib.stream_ptr()), 'hn_variance_to_inv_s')
            var = 0.0                                   # (not read by the single-field renders of such a field)
            self._variance, self._inv_s = None, None
        else:
            var = float(variance.detach().cpu()) if isinstance(variance, torch.Tensor) else float(variance)
        if scale is None:
            scale = float(getattr(sdf, 'scale', 1.0)) if not isinstance(sdf, dict) else 1.0
        keep = []
        d_sdf, d_col = _mlp_desc(sdf_sd, keep), _mlp_desc(col_sd, keep)
        handle = ctypes.c_void_p()
        # (a training re-pack -- eval_only, everything on the current stream, whose order keeps the recycled device blocks safe -- waits
        #  for nothing; any other pack first lets the device finish what may still read the blocks it recycles)
        if not (self.eval_only and self.inv_s_t is not None):
            torch.cuda.synchronize()
        rc = self.lib.hn_field_create(_lib.HN_FIELD_OBJ if kind == 'obj' else _lib.HN_FIELD_HAND,
                                      ctypes.byref(d_sdf), ctypes.byref(d_col), var, float(scale),
                                      _lib.PRECISIONS[precision] | (_lib.HN_PACK_EVAL_ONLY if eval_only else 0),
                                      ctypes.byref(handle), _lib.stream_ptr())
        _lib.check(rc, 'hn_field_create')
        del keep
        self.handle = handle
        if self.inv_s_t is not None:
            _lib.check(self.lib.hn_field_set_inv_s_device(handle, _lib.ptr(self.inv_s_t)), 'hn_field_set_inv_s_device')
        else:
            self._variance = var
            self._inv_s = float(self.lib.hn_field_inv_s(handle))

    @property
    def variance(self):
        if self._variance is None:
            self._variance = float(self._variance_t.cpu())
        return self._variance

    @property
    def inv_s(self):
        if self._inv_s is None:
            self._inv_s = float(self.inv_s_t.cpu())
        return self._inv_s

    def s_val(self, n, device):
        """1 / inv_s as an [n, 1] tensor (the `s_val` of the render dictionaries, utils/renderer.py:236, logged only)."""
        if self.inv_s_t is not None:
            return self.inv_s_t.reciprocal().reshape(1, 1).expand(n, 1)     # (a broadcast view: s_val is logged, never written)
        return torch.full((n, 1), 1.0 / self.inv_s, device=device)

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h is not None and h.value:
            try:
                if not (getattr(self, 'eval_only', False) and getattr(self, 'inv_s_t', None) is not None):
                    torch.cuda.synchronize()   # (a training re-pack's predecessor: same stream, its blocks are recycled in stream order)
                self.lib.hn_field_destroy(h)
            except Exception:
                pass
            self.handle = None

    def set_culling(self, enabled):
        """Exact far-field early-out of the hand field (hn_field_set_culling); results stay bit-identical."""
        torch.cuda.synchronize()
        _lib.check(self.lib.hn_field_set_culling(self.handle, 1 if enabled else 0), 'hn_field_set_culling')

    def set_compaction(self, enabled):
        """Exact far-field skip in the two-field renders (hn_field_set_compaction); results stay bit-identical."""
        _lib.check(self.lib.hn_field_set_compaction(self.handle, 1 if enabled else 0), 'hn_field_set_compaction')
        self.compaction = bool(enabled)

    # ---- direct field queries (utils/fields.py .sdf / forward+gradient+colour) ----------
    def _frames(self, pts, bt_inv, T_pose):
        n = pts.shape[0]
        if self.kind == 'obj':
            return None, None, 1, max(n, 1)
        bt = _lib.f32(bt_inv).reshape(-1, 21, 4, 4)
        tp = _lib.f32(T_pose).reshape(-1, 21, 3)
        nf = bt.shape[0]
        if tp.shape[0] != nf:
            tp = tp.expand(nf, 21, 3).contiguous()
        assert n % nf == 0, 'points must split evenly over frames'
        return bt, tp, nf, max(n // nf, 1)

    def sdf(self, pts, bt_inv=None, T_pose=None):
        """[..,3] -> [M,1] (utils/fields.py:158-160, 330-331)."""
        pts = _lib.f32(pts).reshape(-1, 3)
        n = pts.shape[0]
        bt, tp, nf, ppf = self._frames(pts, bt_inv, T_pose)
        out = torch.empty(n, device=pts.device, dtype=torch.float32)
        ws_bytes = self.lib.hn_field_workspace_bytes(self.handle, n)
        ws = torch.empty(max(ws_bytes, 16), device=pts.device, dtype=torch.uint8)
        rc = self.lib.hn_field_sdf(self.handle, _lib.ptr(pts), n, _lib.ptr(bt), _lib.ptr(tp), nf, ppf, _lib.ptr(out),
                                   _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, 'hn_field_sdf')
        return out.reshape(n, 1)

    def evaluate(self, pts, rays_d, samples_per_ray, bt_inv=None, T_pose=None, want_feat=False):
        """pts [n,3], rays_d [n/samples_per_ray,3] -> sdf [n,1], grad [n,3], rgb [n,3] (, feat [n,256])."""
        pts = _lib.f32(pts).reshape(-1, 3)
        rays_d = _lib.f32(rays_d).reshape(-1, 3)
        n = pts.shape[0]
        bt, tp, nf, ppf = self._frames(pts, bt_inv, T_pose)
        dev = pts.device
        sdf = torch.empty(n, device=dev)
        grad = torch.empty(n, 3, device=dev)
        rgb = torch.empty(n, 3, device=dev)
        feat = torch.empty(n, 256, device=dev) if want_feat else None
        ws_bytes = self.lib.hn_field_workspace_bytes(self.handle, n)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        rc = self.lib.hn_field_eval(self.handle, _lib.ptr(pts), _lib.ptr(rays_d), n, int(samples_per_ray),
                                    _lib.ptr(bt), _lib.ptr(tp), nf, ppf, _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb),
                                    _lib.ptr(feat), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, 'hn_field_eval')
        out = (sdf.reshape(n, 1), grad, rgb)
        return out + (feat,) if want_feat else out


_SUBMODULES = weakref.WeakKeyDictionary()     # module -> its submodules (the tree is walked once per module, not at every render)


def params_version(*modules):
    """A cheap fingerprint of the parameters' in-place versions: the adapters
    re-pack when a checkpoint is loaded after construction.  Called several times per optimisation step of the fitting loops
    (every render, the stable term): the parameters are read straight from the submodules' own tables -- `Module.parameters()` walks the
    module tree through generators and de-duplicating sets at every call, 0.17 ms per call for the six networks, 0.8 ms per window step."""
    v = []
    for m in modules:
        if isinstance(m, nn.Module):
            ent = _SUBMODULES.get(m)
            if ent is None or ent[1] != sum(len(sm._modules) for sm in ent[0]):      # (a child added or removed since: walk again)
                subs = list(m.modules())
                ent = _SUBMODULES[m] = (subs, sum(len(sm._modules) for sm in subs))
            for sm in ent[0]:
                for q in sm._parameters.values():
                    if q is not None:
                        v.append((id(q), q._version))
        elif isinstance(m, dict):
            v.append(id(m))
        else:
            v.append(repr(m))
    return tuple(v)

"""Drop-in adapters with the reference's renderer call surface.

`NeuSRenderer` mirrors utils/renderer.py:39-284 and `NeuSRenderer_fitting`
mirrors utils/renderer.py:286-572: same constructor arguments, same `render`
signature and the same keys / shapes in the returned dict -- but every array
operation runs in libhonerf.so (HIP, gfx950) through the C ABI of
include/honerf.h.  The one addition is the optional keyword `t_rand`
(`[B,1]` uniform numbers in [0,1)): when given it replaces the
`torch.rand([batch_size, 1])` draw of utils/renderer.py:211 so a render is
reproducible; when omitted the draw is made exactly where the reference makes
it.
"""
import numpy as np
import torch

from . import lib as _lib
from .nets import PackedField, params_version


def _t_rand(t_rand, shape, device):
    if t_rand is None:
        return torch.rand(shape, device=device)
    return _lib.f32(t_rand, device).reshape(shape)


class _Workspace:
    """Grow-only device byte buffer reused across calls (no per-call hipMalloc)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self.buf


class NeuSRenderer:
    """Single-field renderer (utils/renderer.py:39-284)."""

    def __init__(self, sdf_network, deviation_network, color_network, model_type, n_samples, n_importance, n_outside,
                 up_sample_steps, perturb):
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.model_type = model_type
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        self.precision = None            # None -> lib.DEFAULT_PRECISION ('f16x3'); 'fp32' selects the exact-fp32 kernels
        self._field = None
        self._version = None
        self._ws = _Workspace()
        self.lib = _lib.load()

    # the renderer keeps references to the modules (utils/renderer.py:50-52); the packed copy is
    # rebuilt lazily whenever their parameters change (checkpoints are loaded after construction)
    def field(self):
        ver = params_version(self.sdf_network, self.color_network, self.deviation_network) + (self.precision,)
        if self._field is None or ver != self._version:
            self._field = PackedField(self.model_type, self.sdf_network, self.color_network, self.deviation_network,
                                      precision=self.precision)
            self._version = ver
        return self._field

    def render(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, verts, Ro, To, index, t_rand=None):
        """utils/renderer.py:190-258.  Returns color_fine [B,3], s_val [B,1], cdf_fine [B,S],
        weight_sum [B,1], weight_max [B,1], gradient_error []."""
        if self.perturb <= 0:
            # the reference only works with perturb > 0 as well (SURVEY B-2)
            raise ValueError('render requires perturb > 0, as the reference does')
        f = self.field()
        lib = self.lib
        rays_o = _lib.f32(rays_o).reshape(-1, 3)
        rays_d = _lib.f32(rays_d).reshape(-1, 3)
        dev = rays_o.device
        B = rays_o.shape[0]
        self.index = index
        st = _lib.stream_ptr()
        if self.model_type == 'obj':
            Ro_, To_ = _lib.f32(Ro, dev).reshape(1, 3, 3), _lib.f32(To, dev).reshape(1, 3)
            o2, d2 = torch.empty_like(rays_o), torch.empty_like(rays_d)
            _lib.check(lib.hn_obj_local_fwd(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(Ro_), _lib.ptr(To_), 1, B,
                                            _lib.ptr(o2), _lib.ptr(d2), st), 'hn_obj_local_fwd')
            rays_o, rays_d = o2, d2
            bt = tp = None
        else:
            bt = _lib.f32(bt_inv, dev).reshape(1, 21, 4, 4)
            tp = _lib.f32(T_pose_21, dev).reshape(1, 21, 3)
        tr = _t_rand(t_rand, (B, 1), dev)
        S = self.n_samples + self.n_importance
        color = torch.empty(B, 3, device=dev)
        cdf = torch.empty(B, S, device=dev)
        wsum = torch.empty(B, 1, device=dev)
        wmax = torch.empty(B, 1, device=dev)
        gerr = torch.empty(1, device=dev)
        z = torch.empty(B, S, device=dev)
        need = lib.hn_render_single_workspace_bytes(f.handle, B, self.n_samples, self.n_importance)
        ws = self._ws.get(need, dev)
        rc = lib.hn_render_single(f.handle, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(tr), B, float(near),
                                  float(far), self.n_samples, self.n_importance, self.up_sample_steps, _lib.ptr(bt),
                                  _lib.ptr(tp), _lib.ptr(color), _lib.ptr(cdf), _lib.ptr(wsum), _lib.ptr(wmax),
                                  _lib.ptr(gerr), _lib.ptr(z), _lib.ptr(ws), ws.numel(), st)
        _lib.check(rc, 'hn_render_single')
        self.last_z_vals = z
        return {
            'color_fine': color,
            's_val': torch.full((B, 1), 1.0 / f.inv_s, device=dev),
            'cdf_fine': cdf,
            'weight_sum': wsum,
            'weight_max': wmax,
            'gradient_error': gerr.reshape(()),
        }

    def sdf(self, pts, bt_inv=None, T_pose_21=None):
        """The SDF grid queries of extract_geometry (utils/renderer.py:260-278) in one launch."""
        return self.field().sdf(pts, bt_inv, T_pose_21)


def _wants_grad(*xs):
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


class NeuSRenderer_fitting:
    """Two-field (hand + object) renderer of the fitting stage (utils/renderer.py:286-572)."""

    batched = False

    def __init__(self, sdf_network_hand, deviation_network_hand, color_network_hand, sdf_network_obj,
                 deviation_network_obj, color_network_obj, n_samples, n_importance, n_outside, up_sample_steps,
                 perturb):
        self.sdf_network_hand = sdf_network_hand
        self.deviation_network_hand = deviation_network_hand
        self.color_network_hand = color_network_hand
        self.sdf_network_obj = sdf_network_obj
        self.deviation_network_obj = deviation_network_obj
        self.color_network_obj = color_network_obj
        self.use_multiple_streams = True
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        self.strict_reference = True     # reproduce SURVEY appendix-B quirks (batched SDF-row gather)
        self.precision = None            # None -> lib.DEFAULT_PRECISION
        self._fields = None
        self._version = None
        self._ws = _Workspace()
        self.lib = _lib.load()

    def fields(self):
        mods = (self.sdf_network_hand, self.color_network_hand, self.deviation_network_hand, self.sdf_network_obj,
                self.color_network_obj, self.deviation_network_obj)
        ver = params_version(*mods) + (self.precision,)
        if self._fields is None or ver != self._version:
            self._fields = (PackedField('hand', mods[0], mods[1], mods[2], precision=self.precision),
                            PackedField('obj', mods[3], mods[4], mods[5], precision=self.precision))
            self._version = ver
        return self._fields

    def _render_raw(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, Ro, To, t_rand):
        """rays [F,P,3] (F = 1 for the unbatched class) -> dict of flat device tensors."""
        hand, obj = self.fields()
        lib = self.lib
        F, P = rays_o.shape[0], rays_o.shape[1]
        dev = rays_o.device
        N = F * P
        S = self.n_samples + 2 * self.n_importance
        bt = _lib.f32(bt_inv, dev).reshape(F, 21, 4, 4)
        tp = _lib.f32(T_pose_21, dev).reshape(-1, 21, 3)
        if tp.shape[0] != F:
            tp = tp.expand(F, 21, 3).contiguous()
        Ro_ = _lib.f32(Ro, dev).reshape(F, 3, 3)
        To_ = _lib.f32(To, dev).reshape(F, 3)
        tr = _t_rand(t_rand, (N, 1), dev)
        out = {
            'color': torch.empty(N, 3, device=dev), 'weight_sum': torch.empty(N, 1, device=dev),
            'sdf_hand': torch.empty(N * S, 1, device=dev), 'sdf_obj': torch.empty(N * S, 1, device=dev),
            'grad_hand': torch.empty(N * S, 3, device=dev), 'grad_obj': torch.empty(N * S, 3, device=dev),
            'gerr': torch.empty(2, device=dev), 'z_vals': torch.empty(N, S, device=dev),
        }
        need = lib.hn_render_dual_workspace_bytes(hand.handle, obj.handle, N, self.n_samples, self.n_importance)
        ws = self._ws.get(need, dev)
        rc = lib.hn_render_dual(hand.handle, obj.handle, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(tr), F, P,
                                float(near), float(far), self.n_samples, self.n_importance, self.up_sample_steps,
                                _lib.ptr(bt), _lib.ptr(tp), _lib.ptr(Ro_), _lib.ptr(To_),
                                1 if (self.strict_reference and F > 1) else 0, _lib.ptr(out['color']),
                                _lib.ptr(out['weight_sum']), _lib.ptr(out['sdf_hand']), _lib.ptr(out['sdf_obj']),
                                _lib.ptr(out['grad_hand']), _lib.ptr(out['grad_obj']), _lib.ptr(out['gerr']),
                                _lib.ptr(out['z_vals']), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, 'hn_render_dual')
        return out

    def render(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, verts, Ro, To, get_SDF=False, t_rand=None):
        """utils/renderer.py:434-535.  rays [B,3]."""
        if self.perturb <= 0:
            raise ValueError('render requires perturb > 0, as the reference does')
        if _wants_grad(rays_o, rays_d, bt_inv, T_pose_21, Ro, To):
            return self._render_autograd(rays_o, rays_d, near, far, bt_inv, T_pose_21, Ro, To, t_rand, (1, -1))
        ro = _lib.f32(rays_o).reshape(1, -1, 3)
        rd = _lib.f32(rays_d).reshape(1, -1, 3)
        o = self._render_raw(ro, rd, near, far, bt_inv, T_pose_21, Ro, To, t_rand)
        self.last_z_vals = o['z_vals']
        return {
            'color_fine': o['color'],
            'weight_sum': o['weight_sum'],
            'sdf_hand': o['sdf_hand'],
            'sdf_obj': o['sdf_obj'],
            'gradient_error_hand': o['gerr'][0],
            'gradient_error_obj': o['gerr'][1],
            'gradient_hand': o['grad_hand'],
            'gradient_obj': o['grad_obj'],
        }

    def _render_autograd(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, Ro, To, t_rand, lead):
        """The same render with the outputs attached to the autograd graph of the pose-dependent inputs
        (what fitting_single.py:289-291 / fitting_video.py:340-342 back-propagate through)."""
        from .autograd import DualRenderFn
        dev = torch.device('cuda')
        g = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(x)).to(device=dev, dtype=torch.float32)
        ro, rd = g(rays_o), g(rays_d)
        if lead == (1, -1):
            ro, rd = ro.reshape(1, -1, 3), rd.reshape(1, -1, 3)
        F, P = ro.shape[0], ro.shape[1]
        outs = DualRenderFn.apply(ro, rd, g(bt_inv), g(T_pose_21), g(Ro), g(To), self, near, far, t_rand)
        color, wsum, sdf_h, sdf_o, grad_h, grad_o, gerr = outs
        if self.batched:
            color, wsum = color.reshape(F, P, 3), wsum.reshape(F, P, 1)
        return {
            'color_fine': color, 'weight_sum': wsum, 'sdf_hand': sdf_h, 'sdf_obj': sdf_o,
            'gradient_error_hand': gerr[0], 'gradient_error_obj': gerr[1], 'gradient_hand': grad_h, 'gradient_obj': grad_o,
        }

    def get_inner_point_id(self, pts, bt_inv, T_pose_21):
        """utils/renderer.py:566-572: indices of points with hand sdf <= 0."""
        val = self.fields()[0].sdf(pts, bt_inv, T_pose_21).detach().cpu().numpy().reshape(-1)
        return np.array(np.where(val <= 0))[0]

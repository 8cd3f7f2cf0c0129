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
import os

import torch

from . import lib as _lib
from .nets import PackedField, params_version


def _t_rand(t_rand, shape, device):
    if t_rand is None:
        return torch.rand(shape, device=device)
    return _lib.f32(t_rand, device).reshape(shape)


# HONERF_POISON_WORKSPACES=1 (tests, tools/ws_poison_check.py): workspaces are filled with 0xff bytes (fp32 NaNs, int -1) before every use
_POISON = os.environ.get('HONERF_POISON_WORKSPACES') == '1'


class _Workspace:
    """Grow-only device byte buffer reused across calls (no per-call hipMalloc)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            if _POISON:
                self.buf.fill_(0xff)
        elif _POISON:
            self.buf.fill_(0xff)      # (every use starts from NaNs: a kernel that reads what the call did not write shows up in its outputs)
        return self.buf


# ---- the stages of a render, one C-ABI call each (shared by the three renderer classes) --------------------------
def _up_sample(lib, z_vals, sdf, n_importance, inv_s):
    """up_sample + sample_pdf(det=True) (utils/renderer.py:60-86, 10-37): z, sdf [R,k] -> [R,n_importance]."""
    z = _lib.f32(z_vals)
    R, k = z.reshape(-1, z.shape[-1]).shape
    sdf = _lib.f32(sdf, z.device).reshape(R, k)
    z_new = torch.empty(R, int(n_importance), device=z.device)
    _lib.check(lib.hn_upsample(_lib.ptr(z.reshape(R, k)), _lib.ptr(sdf), R, k, int(n_importance), float(inv_s), _lib.ptr(z_new),
                               None, _lib.stream_ptr()), 'hn_upsample')
    return z_new.reshape(*z.shape[:-1], int(n_importance))


def _points(lib, rays_o, rays_d, z, mid, sample_dist):
    """p = o + d z (mid = 0) or the section mid-points and dists (mid = 1): utils/renderer.py:216, 119-123."""
    R, k = z.shape
    pts = torch.empty(R * k, 3, device=z.device)
    dists = torch.empty(R * k, device=z.device) if mid else None
    _lib.check(lib.hn_sample_points(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z), R, k, 1 if mid else 0, float(sample_dist),
                                    _lib.ptr(pts), _lib.ptr(dists), _lib.stream_ptr()), 'hn_sample_points')
    return pts, dists


def _cat_z_vals(lib, field, rays_o, rays_d, z_vals, new_z_vals, sdf, bt_inv, T_pose_21, last, quirk_rays_per_frame=0,
                n_frames=1):
    """cat_z_vals (utils/renderer.py:88-105; batched with its row quirk: utils/renderer_batch.py:96-113).
    rays [R,3]; z [R,k]; new z [R,m]; sdf [R,k] -> (z [R,k+m], sdf [R,k+m]); with last=True sdf is returned unchanged,
    as the reference does."""
    R, k = z_vals.shape
    m = new_z_vals.shape[-1]
    z_out = torch.empty(R, k + m, device=z_vals.device)
    st = _lib.stream_ptr()
    if last:
        _lib.check(lib.hn_merge(_lib.ptr(z_vals), _lib.ptr(new_z_vals), None, None, R, k, m, 0, _lib.ptr(z_out), None, None, st),
                   'hn_merge')
        return z_out, sdf
    pts, _ = _points(lib, rays_o, rays_d, new_z_vals, 0, 0.0)
    if field.kind == 'hand':
        new_sdf = field.sdf(pts, _lib.f32(bt_inv).reshape(n_frames, 21, 4, 4), _lib.f32(T_pose_21).reshape(-1, 21, 3))
    else:
        new_sdf = field.sdf(pts)
    new_sdf = new_sdf.reshape(R, m)
    sdf_out = torch.empty(R, k + m, device=z_vals.device)
    _lib.check(lib.hn_merge(_lib.ptr(z_vals), _lib.ptr(new_z_vals), _lib.ptr(sdf), _lib.ptr(new_sdf), R, k, m,
                            int(quirk_rays_per_frame), _lib.ptr(z_out), _lib.ptr(sdf_out), None, st), 'hn_merge')
    return z_out, sdf_out


def _alpha_color(lib, field, rays_o, rays_d, z, sample_dist, bt_inv, T_pose_21, n_frames=1, want_c=False):
    """The module calls + SDF -> alpha stage of render_core / get_alpha_sample_color (utils/renderer.py:119-161,
    369-415) at depths z [R,S]: -> dict(alpha [R,S], c [R,S] | None, rgb [R*S,3], sdf [R*S], grad [R*S,3], dists)."""
    R, S = z.shape
    n = R * S
    dev = z.device
    pts, dists = _points(lib, rays_o, rays_d, z, 1, sample_dist)
    if field.kind == 'hand':
        sdf, grad, rgb = field.evaluate(pts, rays_d, S, _lib.f32(bt_inv).reshape(n_frames, 21, 4, 4),
                                        _lib.f32(T_pose_21).reshape(-1, 21, 3))
    else:
        sdf, grad, rgb = field.evaluate(pts, rays_d, S)
    sdf = sdf.reshape(n)
    alpha = torch.empty(n, device=dev)
    c = torch.empty(n, device=dev) if want_c else None
    _lib.check(lib.hn_alpha(_lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rays_d), _lib.ptr(dists), n, S, float(field.inv_s),
                            _lib.ptr(alpha), _lib.ptr(c), _lib.stream_ptr()), 'hn_alpha')
    return dict(alpha=alpha.reshape(R, S), c=None if c is None else c.reshape(R, S), rgb=rgb, sdf=sdf, grad=grad, dists=dists)


def _grid_points(bound_min, bound_max, resolution, device):
    """The sample grid of extract_geometry (utils/renderer.py:263-273): linspace per axis, 'ij' mesh, x slowest."""
    bmin = torch.as_tensor(bound_min, dtype=torch.float32).reshape(3).cpu()
    bmax = torch.as_tensor(bound_max, dtype=torch.float32).reshape(3).cpu()
    ax = [torch.linspace(float(bmin[i]), float(bmax[i]), int(resolution)) for i in range(3)]
    xx, yy, zz = torch.meshgrid(ax[0], ax[1], ax[2], indexing='ij')
    return torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).to(device).contiguous(), bmin.numpy(), bmax.numpy()


def _marching_cubes(u, threshold, resolution, b_min_np, b_max_np):
    """utils/renderer.py:279-284.  PyMCubes is the reference's third-party dependency; it stays one here."""
    try:
        import mcubes
    except ImportError as e:          # pragma: no cover - not installed in the build image
        raise ImportError('extract_geometry needs PyMCubes (`mcubes`), as the reference does; '
                          'extract_fields() returns the SDF volume without it') from e
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    triangles = triangles[..., ::-1]
    vertices = vertices / (resolution - 1.0) * (b_max_np - b_min_np)[None, :] + b_min_np[None, :]
    return vertices, triangles


def _wants_grad(*xs):
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


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
        # exact far-field skip of the hand field (hn_field_set_compaction): samples whose 21 bone masks are all exactly 0 are not
        # evaluated, bit-identical results.  Off here (throughput is quoted dense); the fitting renderers have it on.
        self.compact_far_field = False
        self._field = None
        self._version = None
        self._ws = _Workspace()
        self.lib = _lib.load()

    # the renderer keeps references to the modules (utils/renderer.py:50-52); the packed copy is
    # rebuilt lazily whenever their parameters change (checkpoints are loaded after construction)
    def mark_parameters_changed(self):
        """Forces a re-pack at the next call.  `field()` notices in-place parameter updates through the tensors' version
        counters; a fused optimiser step (`torch.optim.Adam(fused=True)`) does not advance them, so the training path
        (training.render_train) calls this before every render instead of relying on them."""
        self._version = None

    def field(self):
        eval_only = bool(getattr(self, 'pack_eval_only', False))      # set by training.render_train
        compact = bool(getattr(self, 'compact_far_field', False)) and self.model_type == 'hand'
        ver = params_version(self.sdf_network, self.color_network, self.deviation_network) + (self.precision, eval_only, compact)
        if self._field is None or ver != self._version:
            self._field = PackedField(self.model_type, self.sdf_network, self.color_network, self.deviation_network,
                                      precision=self.precision, eval_only=eval_only, device_variance=True)
            if compact:
                self._field.set_compaction(True)
            self._version = ver
        return self._field

    def _needs_graph(self, *inputs):
        """True when the caller can differentiate this render: grad mode is on and a pose-side input or a parameter of the
        three modules requires grad (exp_runner.train; `--mode test` runs under no_grad or with frozen inputs)."""
        if not torch.is_grad_enabled():
            return False
        if _wants_grad(*inputs):
            return True
        return any(p.requires_grad for m in (self.sdf_network, self.color_network, self.deviation_network)
                   for p in m.parameters())

    def render(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, verts, Ro, To, index, t_rand=None):
        """utils/renderer.py:190-258.  Returns color_fine [B,3], s_val [B,1], cdf_fine [B,S],
        weight_sum [B,1], weight_max [B,1], gradient_error [].  Differentiable as the reference's is (dispatch on
        `_needs_graph`): under `torch.no_grad()` -- every test / validation call site -- it is the plain launch sequence."""
        if self.perturb <= 0:
            # the reference only works with perturb > 0 as well (SURVEY B-2)
            raise ValueError('render requires perturb > 0, as the reference does')
        if self._needs_graph(rays_o, rays_d, bt_inv, T_pose_21, Ro, To):
            # the reference's render carries autograd into the three modules and into whatever Ro / To / bt_inv were
            # computed from (`se3_refine`, exp_runner.py:155-161, 196-232): the same call, attached to the graph
            # (re-packing is left to field()'s version check, as in the plain path: the reference's optimiser is torch's
            # default Adam, whose in-place updates advance the parameters' version counters)
            from .training import render_train
            return render_train(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, verts, Ro, To, index=index, t_rand=t_rand,
                                repack=False, keep_far_field_setting=True)
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
            's_val': f.s_val(B, dev),
            'cdf_fine': cdf,
            'weight_sum': wsum,
            'weight_max': wmax,
            'gradient_error': gerr.reshape(()),
        }

    def sdf(self, pts, bt_inv=None, T_pose_21=None):
        """The SDF grid queries of extract_geometry (utils/renderer.py:260-278) in one launch."""
        return self.field().sdf(pts, bt_inv, T_pose_21)

    # ---- the reference's building blocks, callable on their own as in utils/renderer.py ---------------------------
    def convert_obj_to_local(self, rays_o, rays_d, Ro, To):
        """utils/renderer.py:180-188: o' = Ro (o - To), d' = Ro d."""
        rays_o, rays_d = _lib.f32(rays_o).reshape(-1, 3), _lib.f32(rays_d).reshape(-1, 3)
        dev = rays_o.device
        o2, d2 = torch.empty_like(rays_o), torch.empty_like(rays_d)
        Ro_, To_ = _lib.f32(Ro, dev).reshape(1, 3, 3), _lib.f32(To, dev).reshape(1, 3)   # named: alive until the launch is queued
        _lib.check(self.lib.hn_obj_local_fwd(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(Ro_), _lib.ptr(To_), 1, rays_o.shape[0],
                                             _lib.ptr(o2), _lib.ptr(d2), _lib.stream_ptr()), 'hn_obj_local_fwd')
        return o2, d2

    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        """utils/renderer.py:60-86 -> z_samples [B, n_importance]."""
        return _up_sample(self.lib, z_vals, sdf, n_importance, inv_s)

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, bt_inv, T_pose_21, last=False):
        """utils/renderer.py:88-105 -> (z_vals [B,k+m], sdf [B,k+m])."""
        ro, rd = _lib.f32(rays_o).reshape(-1, 3), _lib.f32(rays_d).reshape(-1, 3)
        return _cat_z_vals(self.lib, self.field(), ro, rd, _lib.f32(z_vals), _lib.f32(new_z_vals),
                           None if sdf is None else _lib.f32(sdf).reshape(z_vals.shape), bt_inv, T_pose_21, last)

    def render_core(self, rays_o, rays_d, bt_inv, T_pose_21, verts, z_vals, sample_dist, sdf_network=None,
                    deviation_network=None, color_network=None):
        """utils/renderer.py:107-177 at the given depths z_vals [B,S] (rays already in the field's frame).  The three
        network arguments are accepted for signature parity; the evaluation uses the renderer's own modules, which is
        what every reference call site passes (utils/renderer.py:237-245).  Returns color [B,3], s_val [B*S,1],
        weights [B,S], cdf [B,S], gradient_error []."""
        for given, own in ((sdf_network, self.sdf_network), (deviation_network, self.deviation_network),
                           (color_network, self.color_network)):
            if given is not None and given is not own:
                raise ValueError('render_core evaluates the networks this renderer was constructed with')
        f = self.field()
        ro, rd = _lib.f32(rays_o).reshape(-1, 3), _lib.f32(rays_d).reshape(-1, 3)
        z = _lib.f32(z_vals, ro.device)
        B, S = z.shape
        dev = ro.device
        a = _alpha_color(self.lib, f, ro, rd, z, sample_dist, bt_inv, T_pose_21, want_c=True)
        color, weights = torch.empty(B, 3, device=dev), torch.empty(B, S, device=dev)
        wsum, wmax, gerr = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.zeros(1, device=dev)
        _lib.check(self.lib.hn_composite1(_lib.ptr(a['alpha']), _lib.ptr(a['c']), _lib.ptr(a['rgb']), _lib.ptr(a['grad']), B, S,
                                          _lib.ptr(color), _lib.ptr(weights), _lib.ptr(wsum), _lib.ptr(wmax), _lib.ptr(gerr),
                                          _lib.stream_ptr()), 'hn_composite1')
        self.N = B * S
        return {
            'color': color,
            's_val': f.s_val(B * S, dev),
            'weights': weights,
            'cdf': a['c'],
            'gradient_error': (gerr / float(B * S)).reshape(()),
        }

    def extract_fields(self, bound_min, bound_max, resolution, bt_inv=None, T_pose_21=None):
        """The SDF volume u [res,res,res] of extract_geometry (utils/renderer.py:260-278): the reference walks the grid
        in 64^3 host chunks, here the whole grid is one hn_field_sdf launch."""
        dev = torch.device('cuda')
        pts, _, _ = _grid_points(bound_min, bound_max, resolution, dev)
        with torch.no_grad():
            val = self.field().sdf(pts, bt_inv, T_pose_21)
        return val.reshape(resolution, resolution, resolution).cpu().numpy()

    def extract_geometry(self, bound_min, bound_max, resolution, bt_inv, T_pose_21, Ro, To, threshold=0.0):
        """utils/renderer.py:260-284 -> (vertices, triangles)."""
        _, bmin, bmax = _grid_points(bound_min, bound_max, 2, torch.device('cpu'))
        u = self.extract_fields(bound_min, bound_max, resolution, bt_inv, T_pose_21)
        return _marching_cubes(u, threshold, resolution, bmin, bmax)


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
        # exact far-field skip of the hand field in the two-field renders (hn_field_set_compaction): samples whose 21 bone
        # masks are all exactly 0 are not evaluated (bit-identical results, 40 - 60 % fewer samples in a fitting step).
        # Set it before the first render; False evaluates every sample ("dense").
        self.compact_far_field = True
        self._fields = None
        self._version = None
        self._ws = _Workspace()
        self._ws_bwd = _Workspace()      # workspace of the adjoint launches (autograd.DualRenderFn.backward)
        self._tape = _Workspace()        # tape of the last differentiable render's final evaluation
        self.lib = _lib.load()

    def fields(self):
        mods = (self.sdf_network_hand, self.color_network_hand, self.deviation_network_hand, self.sdf_network_obj,
                self.color_network_obj, self.deviation_network_obj)
        ver = params_version(*mods) + (self.precision, bool(self.compact_far_field))
        if self._fields is None or ver != self._version:
            self._fields = (PackedField('hand', mods[0], mods[1], mods[2], precision=self.precision),
                            PackedField('obj', mods[3], mods[4], mods[5], precision=self.precision))
            self._fields[0].set_compaction(bool(self.compact_far_field))
            self._version = ver
        return self._fields

    def _render_raw(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, Ro, To, t_rand, keep_tape=False):
        """rays [F,P,3] (F = 1 for the unbatched class) -> dict of flat device tensors.  keep_tape: a backward pass will
        follow -- the final evaluation keeps its tape in self._tape for hn_render_dual_bwd."""
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
        tape, tape_bytes = None, 0
        if keep_tape:
            tape_bytes = lib.hn_render_dual_tape_bytes(hand.handle, obj.handle, N, S)
            tape = self._tape.get(tape_bytes, dev) if tape_bytes else None
        out['tape'] = tape
        rc = lib.hn_render_dual(hand.handle, obj.handle, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(tr), F, P,
                                float(near), float(far), self.n_samples, self.n_importance, self.up_sample_steps,
                                _lib.ptr(bt), _lib.ptr(tp), _lib.ptr(Ro_), _lib.ptr(To_),
                                1 if (self.strict_reference and F > 1) else 0, _lib.ptr(out['color']),
                                _lib.ptr(out['weight_sum']), _lib.ptr(out['sdf_hand']), _lib.ptr(out['sdf_obj']),
                                _lib.ptr(out['grad_hand']), _lib.ptr(out['grad_obj']), _lib.ptr(out['gerr']),
                                _lib.ptr(out['z_vals']), _lib.ptr(ws), ws.numel(), _lib.ptr(tape), tape_bytes, 0, _lib.stream_ptr())
        _lib.check(rc, 'hn_render_dual')
        self._last_z_raw = out['z_vals']
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
        self.last_z_vals = self._last_z_raw.reshape(F, P, -1) if self.batched else self._last_z_raw
        if self.batched:
            color, wsum = color.reshape(F, P, 3), wsum.reshape(F, P, 1)
        return {
            'color_fine': color, 'weight_sum': wsum, 'sdf_hand': sdf_h, 'sdf_obj': sdf_o,
            'gradient_error_hand': gerr[0], 'gradient_error_obj': gerr[1], 'gradient_hand': grad_h, 'gradient_obj': grad_o,
        }

    # ---- the reference's building blocks, callable on their own as in utils/renderer.py:313-432 --------------------
    def _field_of(self, ctype):
        return self.fields()[1 if ctype == 'obj' else 0]

    def convert_obj_to_local(self, rays_o, rays_d, Ro, To):
        """utils/renderer.py:424-432 (batched: utils/renderer_batch.py:176-182): o' = Ro (o - To), d' = Ro d."""
        ro, rd = _lib.f32(rays_o), _lib.f32(rays_d)
        shape = ro.shape
        F = shape[0] if self.batched else 1
        P = ro.reshape(F, -1, 3).shape[1]
        dev = ro.device
        o2, d2 = torch.empty(F * P, 3, device=dev), torch.empty(F * P, 3, device=dev)
        Ro_, To_ = _lib.f32(Ro, dev).reshape(F, 3, 3), _lib.f32(To, dev).reshape(F, 3)   # named: alive until the launch is queued
        _lib.check(self.lib.hn_obj_local_fwd(_lib.ptr(ro.reshape(-1, 3)), _lib.ptr(rd.reshape(-1, 3)), _lib.ptr(Ro_), _lib.ptr(To_),
                                             F, P, _lib.ptr(o2), _lib.ptr(d2), _lib.stream_ptr()), 'hn_obj_local_fwd')
        return o2.reshape(shape), d2.reshape(shape)

    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        """utils/renderer.py:313-338 (batched: utils/renderer_batch.py:68-94) -> z_samples [.., n_importance]."""
        return _up_sample(self.lib, z_vals, sdf, n_importance, inv_s)

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, bt_inv, T_pose_21, ctype, last=False):
        """utils/renderer.py:340-357; batched (utils/renderer_batch.py:96-113) with its SDF-row quirk (SURVEY B-1) when
        strict_reference is set.  Shapes follow the class: [B,k] or [F,P,k]."""
        z = _lib.f32(z_vals)
        lead = z.shape[:-1]
        F = lead[0] if self.batched else 1
        R = z.reshape(-1, z.shape[-1]).shape[0]
        ro, rd = _lib.f32(rays_o).reshape(R, 3), _lib.f32(rays_d).reshape(R, 3)
        zn = _lib.f32(new_z_vals).reshape(R, -1)
        quirk = (R // F) if (self.batched and self.strict_reference and F > 1) else 0
        z_out, sdf_out = _cat_z_vals(self.lib, self._field_of(ctype), ro, rd, z.reshape(R, -1), zn,
                                     None if sdf is None else _lib.f32(sdf).reshape(R, -1), bt_inv, T_pose_21, last,
                                     quirk_rays_per_frame=quirk, n_frames=F)
        z_out = z_out.reshape(*lead, -1)
        return z_out, (sdf if last else sdf_out.reshape(*lead, -1))

    def get_alpha_sample_color(self, rays_o, rays_d, bt_inv, T_pose_21, z_vals, sample_dist, ctype, get_SDF=False):
        """utils/renderer.py:360-422 (batched: utils/renderer_batch.py:115-174): one field at the shared depths ->
        (alpha [B,S], sampled_color [B,S,3], sdf [B*S,1], gradient_error [], gradients [B*S,3]); rays in that field's
        frame (object-local for ctype 'obj')."""
        z = _lib.f32(z_vals)
        lead = z.shape[:-1]
        S = z.shape[-1]
        F = lead[0] if self.batched else 1
        R = z.reshape(-1, S).shape[0]
        ro, rd = _lib.f32(rays_o).reshape(R, 3), _lib.f32(rays_d).reshape(R, 3)
        a = _alpha_color(self.lib, self._field_of(ctype), ro, rd, z.reshape(R, S), sample_dist, bt_inv, T_pose_21, n_frames=F)
        nrm = a['grad'].norm(dim=-1)
        return (a['alpha'].reshape(*lead, S), a['rgb'].reshape(*lead, S, 3), a['sdf'].reshape(-1, 1),
                ((nrm - 1.0) ** 2).mean(), a['grad'])

    def extract_fields(self, bound_min, bound_max, resolution, bt_inv, T_pose_21, Ro, To, get_type):
        """The SDF volume of extract_geometry (utils/renderer.py:537-556) in one launch; 'obj' queries go through
        o' = Ro (p - To) first (:550-552)."""
        dev = torch.device('cuda')
        pts, _, _ = _grid_points(bound_min, bound_max, resolution, dev)
        hand, obj = self.fields()
        with torch.no_grad():
            if get_type == 'hand':
                val = hand.sdf(pts, _lib.f32(bt_inv).reshape(-1, 21, 4, 4)[:1], _lib.f32(T_pose_21).reshape(-1, 21, 3)[:1])
            else:
                local = torch.empty_like(pts)
                dummy = torch.empty_like(pts)
                Ro_, To_ = _lib.f32(Ro, dev).reshape(-1, 3, 3)[:1].contiguous(), _lib.f32(To, dev).reshape(-1, 3)[:1].contiguous()
                _lib.check(self.lib.hn_obj_local_fwd(_lib.ptr(pts), _lib.ptr(pts), _lib.ptr(Ro_), _lib.ptr(To_), 1, pts.shape[0],
                                                     _lib.ptr(local), _lib.ptr(dummy), _lib.stream_ptr()), 'hn_obj_local_fwd')
                val = obj.sdf(local)
        return val.reshape(resolution, resolution, resolution).cpu().numpy()

    def extract_geometry(self, bound_min, bound_max, resolution, bt_inv, T_pose_21, Ro, To, get_type, threshold=0.0):
        """utils/renderer.py:537-564 -> (vertices, triangles)."""
        _, bmin, bmax = _grid_points(bound_min, bound_max, 2, torch.device('cpu'))
        u = self.extract_fields(bound_min, bound_max, resolution, bt_inv, T_pose_21, Ro, To, get_type)
        return _marching_cubes(u, threshold, resolution, bmin, bmax)

    def get_inner_point_id(self, pts, bt_inv, T_pose_21):
        """utils/renderer.py:566-572: indices of points with hand sdf <= 0."""
        val = self.fields()[0].sdf(pts, bt_inv, T_pose_21).detach().cpu().numpy().reshape(-1)
        return np.array(np.where(val <= 0))[0]

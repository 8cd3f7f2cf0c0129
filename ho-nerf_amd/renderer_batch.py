"""Frame-batched two-field renderer with the call surface of
utils/renderer_batch.py:41-313 (used by fitting_video.py): rays `[F,P,3]`,
`bt_inv [F,21,4,4]`, `T_pose_21 [F,21,3]`, `Ro [F,3,3]`, `To [F,3]`; colour and
weight outputs keep the leading frame dimension."""
import torch

from . import lib as _lib
from .renderer import NeuSRenderer_fitting as _Unbatched


class NeuSRenderer_fitting(_Unbatched):
    batched = True

    def render(self, rays_o, rays_d, near, far, bt_inv, T_pose_21, verts, Ro, To, get_SDF=False, t_rand=None):
        """utils/renderer_batch.py:184-281."""
        if self.perturb <= 0:
            raise ValueError('render requires perturb > 0, as the reference does')
        from .renderer import _wants_grad
        if _wants_grad(rays_o, rays_d, bt_inv, T_pose_21, Ro, To):
            self.batch_size, self.pixel_sample = rays_o.shape[0], rays_o.shape[1]
            return self._render_autograd(rays_o, rays_d, near, far, bt_inv, T_pose_21, Ro, To, t_rand, None)
        ro = _lib.f32(rays_o)
        rd = _lib.f32(rays_d)
        F, P = ro.shape[0], ro.shape[1]
        self.batch_size, self.pixel_sample = F, P
        o = self._render_raw(ro, rd, near, far, bt_inv, T_pose_21, Ro, To, t_rand)
        self.last_z_vals = o['z_vals'].reshape(F, P, -1)
        return {
            'color_fine': o['color'].reshape(F, P, 3),
            'weight_sum': o['weight_sum'].reshape(F, P, 1),
            'sdf_hand': o['sdf_hand'],
            'sdf_obj': o['sdf_obj'],
            'gradient_error_hand': o['gerr'][0],
            'gradient_error_obj': o['gerr'][1],
            'gradient_hand': o['grad_hand'],
            'gradient_obj': o['grad_obj'],
        }

    STABLE_MAX_FRAMES, STABLE_MAX_SEL = 8, 1024     # hn_stable_value: frames per window, selected vertices (every 10th) per frame

    def fused_stable_applies(self, pts):
        """Whether the fused form of the stable term (StableTerm / StableLossFn over hn_stable_value) takes these vertices: an f16x3
        hand field, at most 8 frames, at most 1024 SELECTED vertices per frame (hn_stable_value keeps the selection of a window in
        LDS) -- i.e. object meshes of up to 10 240 vertices.  Anything else goes through the torch-operator form below, which has no
        limit, as the reference (utils/renderer_batch.py:318-371)."""
        hand = self.fields()[0]
        shape = pts.shape if hasattr(pts, 'shape') else torch.as_tensor(pts).shape
        return (getattr(self, 'fused_stable', True) and (hand.precision or 'f16x3') == 'f16x3' and len(shape) == 3
                and shape[0] <= self.STABLE_MAX_FRAMES and (shape[1] + 9) // 10 <= self.STABLE_MAX_SEL)

    def get_stable_loss_cross(self, pts, bt_inv, T_pose_21, Ro, To, as_term=False):
        """utils/renderer_batch.py:318-371: the 'stable' term of fitting_video (fit type '1234', weight x100 at
        fitting_video.py:322-324).  pts [F,V,3]: the object's vertices per frame of the window; every 10th vertex is
        taken to the world with (Ro, To), the hand SDF is evaluated there, and over the frames in which the hand
        penetrates the object a vertex that is inside the hand in one frame is pushed to be inside in all of them
        (in_err), while the nearest outside vertex of every inside vertex (the reference's cKDTree query on the CPU;
        here hn_nearest_masked on the device) is kept outside (out_err).  Differentiable w.r.t. bt_inv, Ro, To.

        Everything stays on the device and nothing synchronises; the reference's `if len(in_id_list) > 1` becomes a
        `where`.  strict_reference reproduces how the reference forms the 'outside' set: `np.setdiff1d(range(V), mask)`
        is applied to the boolean MASK, not to indices, so what is removed from the vertex list are the integer values
        the mask takes (vertex 1 if any vertex is inside, vertex 0 if any is outside) and the 'outside' candidates
        include the inside vertices themselves; with strict_reference = False the complement of the inside set is used."""
        hand = self.fields()[0]
        if self.fused_stable_applies(pts):
            # the whole term as one autograd node over a handful of launches (autograd.StableLossFn)
            from .autograd import StableLossFn, StableTerm
            from .renderer import _Workspace
            if not hasattr(self, '_stable_state'):
                self._stable_state = {'tape': _Workspace(), 'ws': _Workspace(), 'ws_bwd': _Workspace()}
            g_ = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(x)).to(device='cuda', dtype=torch.float32)
            if as_term:      # the explicit halves (no autograd): the fitting loop's loss node drives the backward pass
                with torch.no_grad():
                    return StableTerm(g_(pts), g_(bt_inv), g_(T_pose_21), g_(Ro), g_(To), hand, self._stable_state, bool(self.strict_reference))
            return StableLossFn.apply(g_(pts), g_(bt_inv), g_(T_pose_21), g_(Ro), g_(To), hand, self._stable_state, bool(self.strict_reference))
        assert not as_term, 'the explicit form of the stable term needs an f16x3 hand field, at most 8 frames and at most 10 240 vertices'
        from .autograd import HandSdfFn
        dev = torch.device('cuda')
        g = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(x)).to(device=dev, dtype=torch.float32)
        pts = g(pts)[:, ::10, :]
        Fr, V, _ = pts.shape
        Ro, To, bt = g(Ro).reshape(Fr, 3, 3), g(To).reshape(Fr, 3), g(bt_inv).reshape(Fr, 21, 4, 4)
        tp = g(T_pose_21).reshape(-1, 21, 3)
        pts_world = (Ro.unsqueeze(1) @ pts.unsqueeze(-1))[..., 0] + To.unsqueeze(1)
        if not hasattr(self, '_ws_stable'):
            from .renderer import _Workspace
            self._ws_stable = _Workspace()       # its own workspace: this term may run on a second stream beside the render
        sdf = HandSdfFn.apply(pts_world.contiguous(), bt, tp, hand, self._ws_stable)          # [F,V]
        with torch.no_grad():
            inside = sdf < 0                                                                    # in_id_list
            pen = inside.any(dim=1)                                                             # frames that penetrate
            in_time = pen.sum().to(torch.float32)
            if self.strict_reference:
                cand = torch.ones(Fr, V, dtype=torch.bool, device=dev)
                cand[:, 1] &= ~inside.any(dim=1)          # the value True (= 1) occurs in the mask
                cand[:, 0] &= ~(~inside).any(dim=1)       # the value False (= 0) occurs in the mask
            else:
                cand = ~inside
            query = (inside & pen[:, None]).to(torch.uint8).contiguous()
            candm = cand.to(torch.uint8).contiguous()
            selected = torch.empty(Fr, V, dtype=torch.uint8, device=dev)
            p0 = _lib.f32(pts[0]).reshape(V, 3)                                                 # the reference queries pts[0]
            _lib.check(self.lib.hn_nearest_masked(_lib.ptr(p0), V, Fr, _lib.ptr(query), _lib.ptr(candm), _lib.ptr(selected), None,
                                                  _lib.stream_ptr()), 'hn_nearest_masked')
            n_in = inside.sum(dim=1).to(torch.float32)
            denom = ((in_time - 1.0) * n_in).clamp_min(1.0)
            w_in = (inside & pen[:, None]).to(torch.float32) / denom[:, None]                   # [cid, vertex]
            w_out = (selected.bool() & pen[:, None]).to(torch.float32) / denom[:, None]
            penf = pen.to(torch.float32)
        pos = (sdf.clip(0, 1e7) * penf[:, None]).sum(0)                                         # over the penetrating frames
        neg = (sdf.clip(-1e7, 0).abs() * penf[:, None]).sum(0)
        total = (w_in.sum(0) * pos).sum() + 0.05 * (w_out.sum(0) * neg).sum()
        return torch.where(in_time > 1, total / in_time.clamp_min(1.0), torch.zeros_like(total))

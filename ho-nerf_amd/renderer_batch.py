"""Frame-batched two-field renderer with the call surface of
utils/renderer_batch.py:41-313 (used by fitting_video.py): rays `[F,P,3]`,
`bt_inv [F,21,4,4]`, `T_pose_21 [F,21,3]`, `Ro [F,3,3]`, `To [F,3]`; colour and
weight outputs keep the leading frame dimension."""
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

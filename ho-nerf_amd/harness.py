"""The image-rendering caller of the renderers: `Runner.test` of exp_runner.py:308-374 restated over the C ABI.

What the reference does per test view: build the full NDC pixel grid (exp_runner.py:338-350), turn it into
rays (`_xy_to_ray_bundle`, utils/utils.py:31-115), split the rays into `batch_size` chunks, call
`renderer.render` per chunk, concatenate `color_fine`, and form the image as `(rgb * 255).clip(0, 255)`
reshaped `[H, W, 3]` (exp_runner.py:356-372).  Here the ray bundle comes from `hn_ray_gen` and one `render`
call covers all H*W rays (chunking stays available for memory-bound hosts through `batch_size`).
"""
import numpy as np
import torch

from . import lib as L
from . import synth


def image_rays(camera, H, W, device):
    """rays_o, rays_d [H*W, 3] of the full pixel grid for one camera dict {'R','T','focal','principal'}
    (arrays shaped like PerspectiveCameras' arguments: [1,3,3], [1,3], [1,2], [1,2])."""
    lib = L.load()
    xy = torch.from_numpy(synth.ndc_grid(H, W)).to(device).contiguous()
    t = {k: torch.as_tensor(np.asarray(camera[k], dtype=np.float32)).to(device).contiguous() for k in ('R', 'T', 'focal', 'principal')}
    B = H * W
    rays_o = torch.empty(B, 3, device=device)
    rays_d = torch.empty(B, 3, device=device)
    L.check(lib.hn_ray_gen(L.ptr(xy), L.ptr(t['R']), L.ptr(t['T']), L.ptr(t['focal']), L.ptr(t['principal']), 1, B,
                           L.ptr(rays_o), L.ptr(rays_d), L.stream_ptr()), 'hn_ray_gen')
    return rays_o, rays_d


def to_image(color_fine, H, W):
    """exp_runner.py:370: `(rgb.reshape(H, W, 3) * 255).clip(0, 255)`, as uint8 like the cv2.imwrite that follows."""
    img = (color_fine.detach().float().cpu().numpy().reshape(H, W, 3) * 255.0).clip(0, 255)
    return img.astype(np.uint8)


def render_image(renderer, camera, H, W, near, far, bt_inv, T_pose_21, Ro=None, To=None, batch_size=None, t_rand=None,
                 index=0):
    """One test view -> uint8 image [H, W, 3] (+ the raw render outputs of the last chunk's keys, concatenated).

    `Ro`/`To` default to the identity pose; as in the reference the renderer receives `Ro.T` (exp_runner.py:365)."""
    device = torch.device('cuda')
    rays_o, rays_d = image_rays(camera, H, W, device)
    Ro = torch.eye(3, device=device) if Ro is None else torch.as_tensor(Ro, dtype=torch.float32, device=device)
    To = torch.zeros(3, device=device) if To is None else torch.as_tensor(To, dtype=torch.float32, device=device)
    B = H * W
    step = B if not batch_size else int(batch_size)
    outs = []
    for s in range(0, B, step):
        kw = {} if t_rand is None else {'t_rand': t_rand[s:s + step]}
        outs.append(renderer.render(rays_o[s:s + step], rays_d[s:s + step], near, far, bt_inv, T_pose_21, None,
                                    Ro.T.contiguous(), To, index, **kw))
    merged = {k: torch.cat([o[k] for o in outs], 0) for k in ('color_fine', 'weight_sum', 'weight_max')}
    return to_image(merged['color_fine'], H, W), merged

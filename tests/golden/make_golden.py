#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE (read-only at
/root/reference) on CPU.  Run in the build container only:

    python tests/golden/make_golden.py

The reference never travels to the GPU box; these vectors do.  Network weights
are NOT stored: they come from honerf_amd.synth.synth_state_dict(kind, seed)
(numpy RandomState, reproducible anywhere) and are loaded into the reference
modules through their own load_state_dict, so a fixture is just inputs +
reference outputs.

Unused-on-the-path imports of the reference (torchvision for VGGLoss, mcubes
for marching cubes) are absent from this image and are stubbed as empty
modules; every function on the rendering path runs unmodified.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('HONERF_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
for name in ('torchvision', 'mcubes'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, REF)

from honerf_amd import synth  # noqa: E402
import utils.fields as rf      # noqa: E402  (reference)
import utils.renderer as rr    # noqa: E402  (reference)
import utils.renderer_batch as rb  # noqa: E402  (reference)

torch.set_num_threads(8)

SEEDS = {'sdf_obj': 11, 'color_obj': 12, 'sdf_hand': 21, 'color_hand': 22}
VAR_OBJ, VAR_HAND = 0.3, 0.27

SDF_OBJ_CONF = dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], v_multires=10, r_multires=4,
                    bias=0.5, scale=1.0, geometric_init=True, weight_norm=True)
SDF_HAND_CONF = dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], v_multires=10, r_multires=7,
                     bias=0.5, scale=1.0, geometric_init=True, weight_norm=True)
COL_OBJ_CONF = dict(d_feature=256, d_in=3, d_out=3, d_hidden=256, n_layers=4, weight_norm=True, v_multires=10,
                    r_multires=4, grad_multires=4, squeeze_out=True, use_gradients=True)
COL_HAND_CONF = dict(d_feature=256, d_in=3, d_out=3, d_hidden=256, n_layers=4, weight_norm=True, v_multires=10,
                     r_multires=7, grad_multires=4, squeeze_out=True, use_gradients=True)


def load(module, kind):
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(kind, SEEDS[kind]).items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m == 'se3_refine' for m in missing), missing
    return module


def build_nets(use_batch=False):
    emb = rf.Embedding()
    nets = {
        'sdf_obj': load(rf.SDFNetwork_OBJ(emb, 1, 'real', **SDF_OBJ_CONF), 'sdf_obj'),
        'color_obj': load(rf.RenderingNetwork_OBJ(emb, 'real', **COL_OBJ_CONF), 'color_obj'),
        'sdf_hand': load(rf.SDFNetwork(emb, 1, 'real', use_batch=use_batch, **SDF_HAND_CONF), 'sdf_hand'),
        'color_hand': load(rf.RenderingNetwork(emb, 'real', **COL_HAND_CONF), 'color_hand'),
        'var_obj': rf.SingleVarianceNetwork(VAR_OBJ),
        'var_hand': rf.SingleVarianceNetwork(VAR_HAND),
    }
    return emb, nets


class Recorder:
    """Records the integer outputs of torch.searchsorted / torch.sort while the
    reference runs (the sample indices that must match bit-exactly)."""

    def __init__(self):
        self.inds, self.index = [], []

    def __enter__(self):
        self._ss, self._sort = torch.searchsorted, torch.sort

        def ss(*a, **k):
            out = self._ss(*a, **k)
            self.inds.append(out.clone())
            return out

        def srt(*a, **k):
            out = self._sort(*a, **k)
            self.index.append(out[1].clone())
            return out

        torch.searchsorted, torch.sort = ss, srt
        return self

    def __exit__(self, *exc):
        torch.searchsorted, torch.sort = self._ss, self._sort


class Capture:
    """Wraps a bound method of a reference renderer and records (args, result) of each call,
    e.g. the final depths handed to render_core / get_alpha_sample_color."""

    def __init__(self, obj, name):
        self.calls = []
        fn = getattr(obj, name)

        def wrapped(*a, **k):
            out = fn(*a, **k)
            self.calls.append((a, out))
            return out

        setattr(obj, name, wrapped)


def per_sample_single(ren, nets, kind, o, d, z_vals, sample_dist, bt_inv=None, T_pose=None):
    """The reference's own networks on the mid-points of the reference's final depths:
    per-sample sdf / gradient / colour goldens of render_core (utils/renderer.py:119-142)."""
    dists = torch.cat([z_vals[..., 1:] - z_vals[..., :-1],
                       torch.Tensor([sample_dist]).expand(z_vals[..., :1].shape)], -1)
    mid = z_vals + dists * 0.5
    pts = (o[:, None, :] + d[:, None, :] * mid[..., :, None]).reshape(-1, 3)
    dirs = d[:, None, :].expand(z_vals.shape[0], z_vals.shape[1], 3).reshape(-1, 3)
    if kind == 'obj':
        out = nets['sdf_obj'](pts)
        grad = nets['sdf_obj'].gradient(pts).squeeze()
        rgb = nets['color_obj'](pts, dirs, out[:, 1:], grad, 0)
    else:
        out, feat, r_, h_ = nets['sdf_hand'](pts, bt_inv, T_pose)
        grad = nets['sdf_hand'].gradient(pts, bt_inv, T_pose).squeeze()
        rgb = nets['color_hand'](dirs, feat, out[:, 1:], h_, grad, 0)
    return dict(ps_sdf=out[:, :1], ps_grad=grad, ps_rgb=rgb.reshape(-1, 3))


def np_(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np_(v) for k, v in arrays.items()})
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024.0))


def rays_for(cam, xy):
    """Rays from the build's own restatement of the PyTorch3D convention (the
    reference's ray step is third-party: SURVEY 8c); everything downstream of
    (rays_o, rays_d) is the reference's."""
    sys.path.insert(0, ROOT)
    from oracle.render import rays_from_xy
    t = lambda a: torch.from_numpy(np.asarray(a))
    return rays_from_xy(t(xy), t(cam['R'][0]), t(cam['T'][0]), t(cam['focal'][0]), t(cam['principal'][0]))


def hand_scene_rays(n, seed):
    """n rays through the synthetic hand (joints near z ~ 0.9-1.07)."""
    bt_inv, T_pose, joints = synth.synth_hand_pose(seed)
    rng = np.random.RandomState(seed + 100)
    cam = synth.front_camera(dist=0.0, focal=2.0)
    # aim at random joints with a little scatter -> rays pass through the bone masks
    tgt = joints[rng.randint(0, 21, size=n)] + 0.012 * rng.standard_normal((n, 3))
    xy = np.stack([tgt[:, 0] / tgt[:, 2] * 2.0, tgt[:, 1] / tgt[:, 2] * 2.0], -1).astype(np.float32)
    o, d = rays_for(cam, xy)
    return o, d, torch.from_numpy(bt_inv), torch.from_numpy(T_pose), joints


def main():
    emb, nets = build_nets()
    g = torch.Generator().manual_seed(1234)

    # ---- a4: Embedding ---------------------------------------------------------------
    x = torch.randn(7, 3, generator=g) * 0.7
    save('embed', x=x, L10=emb(x, 10), L4=emb(x, 4), L7=emb(x, 7))

    # ---- a7: per-bone coordinates ----------------------------------------------------
    o, d, bt_inv, T_pose, joints = hand_scene_rays(12, 3)
    pts = o + d * torch.linspace(0.85, 1.1, 12)[:, None]
    v, r, h = rf.anerf_emb_point(pts, bt_inv, T_pose)
    bt2, T2, _ = synth.synth_hand_pose(4)
    bt_b = torch.stack([bt_inv, torch.from_numpy(bt2)])
    T_b = torch.stack([T_pose, torch.from_numpy(T2)])
    pts_b = torch.stack([pts, pts + 0.01])
    vb, rb_, hb = rf.anerf_emb_point_batch(pts_b, bt_b, T_b)
    save('bone_coords', pts=pts, bt_inv=bt_inv, T_pose=T_pose, v=v, r=r, h=h,
         pts_b=pts_b, bt_inv_b=bt_b, T_pose_b=T_b, v_b=vb, r_b=rb_, h_b=hb)

    # ---- a5/a6: obj field ------------------------------------------------------------
    po = (torch.rand(48, 3, generator=g) - 0.5) * 1.4
    do = torch.nn.functional.normalize(torch.randn(48, 3, generator=g), dim=-1)
    out = nets['sdf_obj'](po)
    grad = nets['sdf_obj'].gradient(po).squeeze()
    rgb = nets['color_obj'](po, do, out[:, 1:], grad, 0)
    save('field_obj', pts=po.detach(), dirs=do, out=out, grad=grad, rgb=rgb)

    # ---- a8/a9: hand field -----------------------------------------------------------
    o, d, bt_inv, T_pose, joints = hand_scene_rays(40, 5)
    ph = o + d * (0.82 + 0.3 * torch.rand(40, 1, generator=g))
    out, feat, r_, h_ = nets['sdf_hand'](ph, bt_inv, T_pose)
    grad = nets['sdf_hand'].gradient(ph, bt_inv, T_pose).squeeze()
    rgb = nets['color_hand'](d, feat, out[:, 1:], h_, grad, 0)
    save('field_hand', pts=ph.detach(), dirs=d, bt_inv=bt_inv, T_pose=T_pose, out=out, feat=feat[:8],
         grad=grad, rgb=rgb, h=h_)

    # ---- a11/a12/a13: up_sample, sample_pdf, cat_z_vals on synthetic z / sdf -----------
    ren = rr.NeuSRenderer(nets['sdf_obj'], nets['var_obj'], nets['color_obj'], 'obj', 64, 64, 0, 4, 1.0)
    B, k = 96, 64
    z = torch.sort(0.4 + 1.1 * torch.rand(B, k, generator=g), dim=-1)[0]
    center = 0.6 + 0.6 * torch.rand(B, 1, generator=g)
    sdf = (z - center).abs() - 0.15 + 0.02 * torch.randn(B, k, generator=g)
    sdf[:8] = 0.3 + 0.05 * torch.randn(8, k, generator=g)         # rays that miss: flat pdf
    steps = {}
    cur_z, cur_sdf = z, sdf
    for i in range(4):
        with Recorder() as rec:
            z_new = ren.up_sample(None, None, cur_z, cur_sdf, 16, 64 * 2 ** i)
        steps['inds%d' % i] = rec.inds[0]
        steps['znew%d' % i] = z_new
        with Recorder() as rec:
            zc = torch.cat([cur_z, z_new], -1)
            zs, index = torch.sort(zc, -1)
        new_sdf = (z_new - center).abs() - 0.15
        sc = torch.cat([cur_sdf, new_sdf], -1)
        ss = torch.gather(sc, 1, index)
        steps['zmerged%d' % i] = zs
        steps['sdfnew%d' % i] = new_sdf
        steps['index%d' % i] = index
        steps['sdfmerged%d' % i] = ss
        cur_z, cur_sdf = zs, ss
    save('upsample', z=z, sdf=sdf, **steps)

    # ---- a14/a15: alpha + single-field compositing through render_core ----------------
    # ---- a17: whole single-field renders ----------------------------------------------
    cam = synth.front_camera(dist=1.0, focal=2.0)
    rng = np.random.RandomState(7)
    xy = (rng.rand(40, 2).astype(np.float32) - 0.5) * 1.2
    o, d = rays_for(cam, xy)
    R_obj, t_obj = synth.synth_obj_pose(2, center=(0.02, -0.01, 0.0))
    Ro = torch.from_numpy(R_obj).T.contiguous()        # callers pass R_obj^T (exp_runner.py:211)
    To = torch.from_numpy(t_obj)
    for tag, nimp in (('obj_64_64', 64), ('obj_32_0', 0)):
        nsamp = 64 if nimp else 32
        ren = rr.NeuSRenderer(nets['sdf_obj'], nets['var_obj'], nets['color_obj'], 'obj', nsamp, nimp, 0, 4, 1.0)
        torch.manual_seed(5)
        t_rand = torch.rand([40, 1])
        torch.manual_seed(5)
        cap = Capture(ren, 'render_core')
        with Recorder() as rec:
            res = ren.render(o, d, 0.4, 1.5, torch.zeros(21, 4, 4), torch.zeros(21, 3), None, Ro, To, 0)
        extra = {('inds%d' % i): t for i, t in enumerate(rec.inds)}
        extra.update({('index%d' % i): t for i, t in enumerate(rec.index)})
        (a, core), = cap.calls
        extra.update(z_vals=a[5], weights=core['weights'])
        extra.update(per_sample_single(ren, nets, 'obj', a[0], a[1], a[5], a[6]))
        save('render_' + tag, rays_o=o, rays_d=d, Ro=Ro, To=To, t_rand=t_rand, near=0.4, far=1.5,
             n_samples=nsamp, n_importance=nimp, **{k: v for k, v in res.items()}, **extra)

    o, d, bt_inv, T_pose, joints = hand_scene_rays(32, 9)
    for tag, nimp in (('hand_64_64', 64), ('hand_64_0', 0)):
        ren = rr.NeuSRenderer(nets['sdf_hand'], nets['var_hand'], nets['color_hand'], 'hand', 64, nimp, 0, 4, 1.0)
        torch.manual_seed(6)
        t_rand = torch.rand([32, 1])
        torch.manual_seed(6)
        cap = Capture(ren, 'render_core')
        with Recorder() as rec:
            res = ren.render(o, d, 0.4, 1.5, bt_inv, T_pose, None, None, None, 0)
        extra = {('inds%d' % i): t for i, t in enumerate(rec.inds)}
        extra.update({('index%d' % i): t for i, t in enumerate(rec.index)})
        (a, core), = cap.calls
        extra.update(z_vals=a[5], weights=core['weights'])
        extra.update(per_sample_single(ren, nets, 'hand', a[0], a[1], a[5], a[6], bt_inv, T_pose))
        save('render_' + tag, rays_o=o, rays_d=d, bt_inv=bt_inv, T_pose=T_pose, t_rand=t_rand, near=0.4,
             far=1.5, n_samples=64, n_importance=nimp, **{k: v for k, v in res.items()}, **extra)

    # ---- a16/a18: two-field render, forward + backward ---------------------------------
    o, d, bt_inv, T_pose, joints = hand_scene_rays(24, 13)
    R_obj, t_obj = synth.synth_obj_pose(3, center=tuple(joints[9] + np.array([0.03, 0.0, 0.02])))
    Ro = torch.from_numpy(R_obj).T.contiguous().requires_grad_(True)
    To = torch.from_numpy(t_obj).clone().requires_grad_(True)
    bt = bt_inv.clone().requires_grad_(True)
    ro = o.clone().requires_grad_(True)
    rd = d.clone().requires_grad_(True)
    ren = rr.NeuSRenderer_fitting(nets['sdf_hand'], nets['var_hand'], nets['color_hand'],
                                  nets['sdf_obj'], nets['var_obj'], nets['color_obj'], 64, 64, 0, 4, 1.0)
    torch.manual_seed(8)
    t_rand = torch.rand([24, 1])
    torch.manual_seed(8)
    cap = Capture(ren, 'get_alpha_sample_color')
    with Recorder() as rec:
        res = ren.render(ro, rd, 0.4, 1.5, bt, T_pose, None, Ro, To)
    (a_h, out_h), (a_o, out_o) = cap.calls
    dual_extra = dict(z_vals=a_h[4], alpha_hand=out_h[0], rgb_hand=out_h[1], alpha_obj=out_o[0], rgb_obj=out_o[1])
    gw = {
        'w_color': torch.randn(24, 3, generator=g), 'w_wsum': torch.randn(24, 1, generator=g),
        'w_sdf_hand': torch.randn(24 * 192, 1, generator=g) * 0.05,
        'w_sdf_obj': torch.randn(24 * 192, 1, generator=g) * 0.05,
    }
    loss = ((res['color_fine'] * gw['w_color']).sum() + (res['weight_sum'] * gw['w_wsum']).sum()
            + (res['sdf_hand'] * gw['w_sdf_hand']).sum() + (res['sdf_obj'] * gw['w_sdf_obj']).sum())
    grads = torch.autograd.grad(loss, [Ro, To, bt, ro, rd])
    save('render_dual', rays_o=o, rays_d=d, bt_inv=bt_inv, T_pose=T_pose, Ro=Ro, To=To, t_rand=t_rand,
         near=0.4, far=1.5, n_samples=64, n_importance=64, **{k: v for k, v in res.items()}, **gw,
         loss=loss, g_Ro=grads[0], g_To=grads[1], g_bt_inv=grads[2], g_rays_o=grads[3], g_rays_d=grads[4],
         inds=torch.stack(rec.inds), **dual_extra)

    # ---- batched two-field render (utils/renderer_batch.py), incl. its SDF-row quirk ----
    emb_b, nets_b = build_nets(use_batch=True)
    Fr, P = 3, 10
    os_, ds_, bts, Ts, Ros, Tos = [], [], [], [], [], []
    for f in range(Fr):
        o, d, bt_inv, T_pose, joints = hand_scene_rays(P, 30 + f)
        R_obj, t_obj = synth.synth_obj_pose(40 + f, center=tuple(joints[9] + np.array([0.03, 0.0, 0.02])))
        os_.append(o); ds_.append(d); bts.append(bt_inv); Ts.append(T_pose)
        Ros.append(torch.from_numpy(R_obj).T.contiguous()); Tos.append(torch.from_numpy(t_obj))
    o, d, bt, Tp, Ro, To = map(torch.stack, (os_, ds_, bts, Ts, Ros, Tos))
    ren = rb.NeuSRenderer_fitting(nets_b['sdf_hand'], nets_b['var_hand'], nets_b['color_hand'],
                                  nets_b['sdf_obj'], nets_b['var_obj'], nets_b['color_obj'], 64, 64, 0, 4, 1.0)
    torch.manual_seed(9)
    t_rand = torch.rand([Fr, P, 1])
    torch.manual_seed(9)
    cap = Capture(ren, 'get_alpha_sample_color')
    res = ren.render(o, d, 0.4, 1.5, bt, Tp, None, Ro, To)
    (a_h, out_h), (a_o, out_o) = cap.calls
    res = dict(res, z_vals=a_h[4], alpha_hand=out_h[0], rgb_hand=out_h[1], alpha_obj=out_o[0], rgb_obj=out_o[1])
    save('render_dual_batch', rays_o=o, rays_d=d, bt_inv=bt, T_pose=Tp, Ro=Ro, To=To, t_rand=t_rand,
         near=0.4, far=1.5, n_samples=64, n_importance=64, **{k: v for k, v in res.items()})


if __name__ == '__main__':
    main()

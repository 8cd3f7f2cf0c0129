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


def reference_block(path, start_marker, end_marker):
    """The statements of a reference entry script between two marker lines (start inclusive, end exclusive),
    dedented, as a code object.  The fitting scripts cannot be imported (pyhocon, cv2, trimesh, pytorch3d and a CUDA
    device are needed at import time), but their loss formulas are plain torch statements inside `Runner.fitting`:
    they are read from the reference file HERE, at fixture-generation time, and executed unmodified on synthetic
    tensors -- nothing of them is stored in this repository, only the numbers they produce."""
    import textwrap
    lines = open(os.path.join(REF, path)).read().split('\n')
    a = next(i for i, l in enumerate(lines) if start_marker in l)
    b = next(i for i, l in enumerate(lines) if end_marker in l and i > a)
    return compile(textwrap.dedent('\n'.join(lines[a:b])), path + ':%d-%d' % (a + 1, b), 'exec')


def loss_goldens(nets, nets_b, g):
    """fitting_single.py:251-288 and fitting_video.py:285-339 executed on renders of the reference renderers:
    loss terms and d loss / d (render outputs, pose-dependent tensors)."""
    import types as _t
    import torch.nn.functional as F_
    quiet = lambda *a, **k: None
    # ---- fitting_single: fit types '1' and '12' on a two-field render --------------------------------------------
    pl_single = {}
    exec(reference_block('fitting_single.py', 'def pose_loss(target_pose, pred_pose)', 'if self.fit_type =='), {'torch': torch}, pl_single)
    blk = reference_block('fitting_single.py', "color_fine = render_out['color_fine']", 'optimizer.zero_grad()')
    o, d, bt_inv, T_pose, joints = hand_scene_rays(24, 13)
    R_obj, t_obj = synth.synth_obj_pose(3, center=tuple(joints[9] + np.array([0.03, 0.0, 0.02])))
    ren = rr.NeuSRenderer_fitting(nets['sdf_hand'], nets['var_hand'], nets['color_hand'],
                                  nets['sdf_obj'], nets['var_obj'], nets['color_obj'], 64, 64, 0, 4, 1.0)
    torch.manual_seed(8)
    res = ren.render(o, d, 0.4, 1.5, bt_inv, T_pose, None, torch.from_numpy(R_obj).T.contiguous(), torch.from_numpy(t_obj))
    # shift the per-sample sdfs so that the contact / penetration selections are non-trivial
    base = {k: res[k].detach().clone() for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj')}
    base['sdf_hand'] = base['sdf_hand'] - base['sdf_hand'].median() - 0.002
    base['sdf_obj'] = (base['sdf_obj'] - base['sdf_obj'].median()) * 0.05 - 0.001
    true_rgb = torch.rand(24, 3, generator=g)
    true_mask = (torch.rand(24, 1, generator=g) > 0.3).float()
    joint3d_pred = torch.from_numpy(joints)
    joint_3d = (joint3d_pred + 0.004 * torch.randn(21, 3, generator=g)).unsqueeze(0)
    obj_verts_loss = torch.tensor(0.0123)
    out = dict(true_rgb=true_rgb, true_mask=true_mask, joint3d_pred=joint3d_pred, joint_3d=joint_3d,
               obj_verts_loss=obj_verts_loss, **{'in_' + k: v for k, v in base.items()})
    for ft in ('1', '12'):
        leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        j3 = joint_3d.clone().requires_grad_(True)
        ns = dict(torch=torch, F=F_, print=quiet, render_out=leaves, true_rgb=true_rgb, true_mask=true_mask,
                  self=_t.SimpleNamespace(fit_type=ft), pose_loss=pl_single['pose_loss'], joint3d_pred=joint3d_pred,
                  joint_3d=j3, joint3d_gt=joint3d_pred, obj_verts_loss=obj_verts_loss, obj_verts_err_to_gt=obj_verts_loss,
                  iter_id=0)
        exec(blk, ns)
        grads = torch.autograd.grad(ns['loss'], [leaves['color_fine'], leaves['weight_sum'], leaves['sdf_hand'], leaves['sdf_obj'], j3],
                                    allow_unused=True)
        z = lambda t_, like: torch.zeros_like(like) if t_ is None else t_
        out.update({'s%s_loss' % ft: ns['loss'], 's%s_color' % ft: ns['color_fine_loss'], 's%s_mask' % ft: ns['mask_loss'],
                    's%s_joint' % ft: ns['joint_loss'],
                    's%s_g_color_fine' % ft: grads[0], 's%s_g_weight_sum' % ft: grads[1],
                    's%s_g_sdf_hand' % ft: z(grads[2], base['sdf_hand']), 's%s_g_sdf_obj' % ft: z(grads[3], base['sdf_obj']),
                    's%s_g_joint_3d' % ft: grads[4]})
        if ft == '12':
            out.update(s12_contact=ns['contact_loss'], s12_penet=ns['penet_loss'])
    save('loss_single', **out)

    # ---- get_stable_loss_cross + the fitting_video losses (fit type '1234') on a 4-frame window -------------------
    Fr, P = 4, 10
    os_, ds_, bts, Ros, Tos, jts = [], [], [], [], [], []
    o0, d0, bt0, T_pose, joints = hand_scene_rays(P, 31)
    rngw = np.random.RandomState(77)
    for f in range(Fr):
        # the same hand, moved rigidly a little from frame to frame; the object sits on the index finger
        Rm = torch.from_numpy(synth._rodrigues(rngw.standard_normal(3), 0.03 * f).astype(np.float32))
        tm = torch.tensor([0.002 * f, -0.001 * f, 0.0015 * f])
        G = torch.eye(4)
        G[:3, :3] = Rm
        G[:3, 3] = tm + torch.from_numpy(joints[0]) - Rm @ torch.from_numpy(joints[0])
        bts.append(bt0 @ torch.inverse(G))
        jts.append((Rm @ (torch.from_numpy(joints) - torch.from_numpy(joints[0])).T).T + torch.from_numpy(joints[0]) + tm)
        R_obj, t_obj = synth.synth_obj_pose(50, center=tuple(joints[6] + np.array([0.004, 0.0, 0.004]) * (1 + 0.3 * f)))
        Ros.append(torch.from_numpy(R_obj)); Tos.append(torch.from_numpy(t_obj))
        o, d, _, _, _ = hand_scene_rays(P, 31 + f)
        os_.append(o); ds_.append(d)
    o, d, bt, obj_r, obj_t, joint_3d = map(torch.stack, (os_, ds_, bts, Ros, Tos, jts))
    Tp = T_pose.unsqueeze(0).expand(Fr, 21, 3).contiguous()
    # object vertices: a 2.2 cm ellipsoid shell in object-local coordinates, 1500 vertices (150 after the [::10])
    u = rngw.standard_normal((1500, 3))
    verts = (u / np.linalg.norm(u, axis=1, keepdims=True) * np.array([0.022, 0.018, 0.02])).astype(np.float32)
    obj_verts = torch.from_numpy(verts).unsqueeze(0).expand(Fr, -1, -1).contiguous()
    renb = rb.NeuSRenderer_fitting(nets_b['sdf_hand'], nets_b['var_hand'], nets_b['color_hand'],
                                   nets_b['sdf_obj'], nets_b['var_obj'], nets_b['color_obj'], 64, 64, 0, 4, 1.0)
    renb.batch_size, renb.pixel_sample = Fr, P
    bt_l, r_l, t_l = bt.clone().requires_grad_(True), obj_r.clone().requires_grad_(True), obj_t.clone().requires_grad_(True)
    with Recorder():
        stable = renb.get_stable_loss_cross(obj_verts, bt_l, Tp, r_l, t_l)
    sg = torch.autograd.grad(stable, [bt_l, r_l, t_l])
    pts_world = (obj_r.unsqueeze(1) @ obj_verts[:, ::10, :].unsqueeze(-1))[..., 0] + obj_t.unsqueeze(1)
    hand_sdf = nets_b['sdf_hand'].sdf(pts_world, bt, Tp).reshape(Fr, -1).detach()
    print('stable loss %.6f; penetrating frames %d; inside per frame %s'
          % (float(stable), int((hand_sdf < 0).any(1).sum()), (hand_sdf < 0).sum(1).tolist()))
    assert float(stable) > 0 and int((hand_sdf < 0).any(1).sum()) >= 2
    save('stable_loss', obj_verts=obj_verts, bt_inv=bt, T_pose=Tp, obj_r=obj_r, obj_t=obj_t, hand_sdf=hand_sdf,
         stable=stable, g_bt_inv=sg[0], g_obj_r=sg[1], g_obj_t=sg[2])

    pl_video = {}
    exec(reference_block('fitting_video.py', 'def pose_loss(target_pose, pred_pose)', 'get_render_all = False'), {'torch': torch}, pl_video)
    blkv = reference_block('fitting_video.py', "color_fine = render_out['color_fine']", 'optimizer.zero_grad()')
    torch.manual_seed(9)
    resv = renb.render(o, d, 0.4, 1.5, bt, Tp, None, torch.inverse(obj_r), obj_t)
    basev = {k: resv[k].detach().clone() for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj')}
    basev['sdf_hand'] = basev['sdf_hand'] - basev['sdf_hand'].median() - 0.002
    basev['sdf_obj'] = (basev['sdf_obj'] - basev['sdf_obj'].median()) * 0.05 - 0.001
    true_rgb = torch.rand(Fr, P, 3, generator=g)
    true_mask = (torch.rand(Fr, P, 1, generator=g) > 0.3).float()
    joint3d_pred = joint_3d + 0.003 * torch.randn(Fr, 21, 3, generator=g)
    pred_v = (obj_r.unsqueeze(1) @ obj_verts[:, ::10].unsqueeze(-1))[..., 0] + obj_t.unsqueeze(1)   # 150 vertices: small fixture
    comp_v = pred_v + 0.002 * torch.randn(pred_v.shape, generator=g)
    outv = dict(true_rgb=true_rgb, true_mask=true_mask, joint3d_pred=joint3d_pred, joint_3d=joint_3d, pred_obj_v_w=pred_v,
                compare_obj_v_w=comp_v, obj_verts_loss=pl_video['pose_loss'](pred_v, comp_v),
                **{'in_' + k: v for k, v in basev.items()})
    for tag, index, first in (('mid', [3, 4, 5, 6], False), ('head', [0, 1, 2, 3], False), ('tail', [6, 7, 8, 9], False),
                              ('first', [0, 1, 2, 3], True)):
        leaves = {k: v.clone().requires_grad_(True) for k, v in basev.items()}
        j3, pv = joint_3d.clone().requires_grad_(True), pred_v.clone().requires_grad_(True)
        ns = dict(torch=torch, F=F_, print=quiet, render_out=leaves, true_rgb=true_rgb, true_mask=true_mask,
                  self=_t.SimpleNamespace(fit_type='1234', renderer=_t.SimpleNamespace(get_stable_loss_cross=lambda *a: stable.detach())),
                  pose_loss=pl_video['pose_loss'], joint3d_pred=joint3d_pred, joint_3d=j3, joint3d_gt=joint3d_pred,
                  pred_obj_v_w=pv, compare_obj_v_w=comp_v, obj_verts_loss=pl_video['pose_loss'](pv, comp_v),
                  obj_verts_err_to_gt=torch.tensor(0.0), iter_id=0, sub_iter_id=0, view_id=0 if first else 1,
                  index=torch.tensor(index), data_num=10, obj_verts=obj_verts, bone_transformation_inv=bt, T_pose_21=Tp,
                  obj_r=obj_r, obj_t=obj_t)
        exec(blkv, ns)
        grads = torch.autograd.grad(ns['loss'], [leaves['color_fine'], leaves['weight_sum'], leaves['sdf_hand'], leaves['sdf_obj'], j3, pv])
        outv.update({tag + '_loss': ns['loss'], tag + '_smooth': ns['smooth_loss'], tag + '_color': ns['color_fine_loss'],
                     tag + '_mask': ns['mask_loss'], tag + '_contact': ns['contact_loss'], tag + '_penet': ns['penet_loss'],
                     tag + '_index': np.asarray(index), tag + '_first': first,
                     tag + '_g_color_fine': grads[0], tag + '_g_weight_sum': grads[1], tag + '_g_sdf_hand': grads[2],
                     tag + '_g_sdf_obj': grads[3], tag + '_g_joint_3d': grads[4], tag + '_g_pred_obj_v_w': grads[5]})
    save('loss_video', stable=stable.detach(), **outv)


def pose_goldens():
    """The hand pose chain of the fitting loops, fitting_single.py:206-226, executed as the reference has it (the
    statements are read from the reference file here and run unmodified; halo_util is imported) on synthetic predicted
    joints: bone_transformation_inv, the refined joints and the full Jacobian w.r.t. the four refine parameters.
    Run in double precision (torch default dtype float64 while the reference's modules are constructed and executed), so
    that the fixture is the chain's exact value; the fp32 run's distance from it is stored beside it as the noise floor."""
    import types as _t
    import torch.nn.functional as F_
    from halo_util.converter_fit_batch import PoseConverter, transform_to_canonical
    from halo_util.utils import convert_joints
    blk = reference_block('fitting_single.py', "kps_local_cs = convert_joints(ori_3d_pose, source='mano', target='biomech')",
                          'obj_rots = rot6d_to_matrix(obj_rot_refine)')
    r6 = reference_block('utils/utils.py', 'def rot6d_to_matrix(rot_6d)', 'def _xy_to_ray_bundle')
    m2b = np.array([0, 1, 5, 9, 13, 17, 2, 6, 10, 14, 18, 3, 7, 11, 15, 19, 4, 8, 12, 16, 20])   # halo_util/utils.py:20

    def run(dtype, ori, bl, prm, want_jac):
        torch.set_default_dtype(dtype)
        try:
            ns0 = {}
            exec(r6, {'torch': torch, 'F': F_}, ns0)
            pc = PoseConverter(dev=torch.device('cpu'))
            ori_t, bl_t = torch.tensor(ori, dtype=dtype).unsqueeze(0), torch.tensor(bl, dtype=dtype).unsqueeze(0)

            def chain(flat):
                ns = dict(torch=torch, convert_joints=convert_joints, transform_to_canonical=transform_to_canonical,
                          rot6d_to_matrix=ns0['rot6d_to_matrix'], self=_t.SimpleNamespace(pose_converter=pc, device=torch.device('cpu')),
                          ori_3d_pose=ori_t, cur_bone_length=bl_t, joint_refine_angle=flat[0:20].unsqueeze(0),
                          palm_refine_angle=flat[20:27].unsqueeze(0), palm_rot_refine=flat[27:33].reshape(1, 3, 2),
                          palm_trans_refine=flat[33:36].unsqueeze(0))
                exec(blk, ns)
                return torch.cat([ns['bone_transformation_inv'].reshape(-1), ns['joint_3d'].reshape(-1)])

            flat = torch.tensor(prm, dtype=dtype)
            out = chain(flat).detach()
            jac = torch.autograd.functional.jacobian(chain, flat).detach() if want_jac else None
            return out.numpy().astype(np.float64), None if jac is None else jac.numpy().astype(np.float64)
        finally:
            torch.set_default_dtype(torch.float32)

    rng = np.random.RandomState(2024)
    N = 6
    oris, bls, prms, outs, jacs, floor = [], [], [], [], [], []
    for c in range(N):
        _, _, joints = synth.synth_hand_pose(100 + c, center=(0.01 * c, -0.02, 0.9), flex=0.3 + 0.05 * c)
        joints = joints.astype(np.float64) + 0.004 * rng.standard_normal((21, 3))       # off the synthetic hand's plane
        kb = joints[m2b]
        bl = np.array([np.linalg.norm(kb[i + 1] - kb[0 if i < 5 else i - 4]) for i in range(20)]) * (1.0 + 0.05 * rng.standard_normal(20))
        prm = np.zeros(36)
        prm[27:33] = np.eye(3)[:, :2].reshape(-1)
        if c > 0:   # case 0: the initial state of the optimisation (all refinements zero)
            prm[0:20] = 0.1 * rng.standard_normal(20)
            prm[20:27] = 0.3 * rng.standard_normal(7)
            prm[27:33] += 0.15 * rng.standard_normal(6)
            prm[33:36] = 0.01 * rng.standard_normal(3)
        o64, j64 = run(torch.float64, joints, bl, prm, True)
        o32, _ = run(torch.float32, joints, bl, prm, False)
        oris.append(joints); bls.append(bl); prms.append(prm); outs.append(o64); jacs.append(j64)
        floor.append(np.abs(o32 - o64).max() / np.abs(o64).max())
        print('pose chain case %d: |out| max %.3f, fp32 run vs fp64 run %.2e, |jac| max %.3f' % (c, np.abs(o64).max(), floor[-1], np.abs(j64).max()))
    # ---- round 5: 24 more hands, so that the chain is pinned WHERE IT IS USED and away from the easy middle.
    # Cases 0 .. 5 above are unchanged (same generator stream).  `kinds` names every case in the fixture.
    kinds = ['initial state'] + ['moderate refinements'] * 5
    rng2 = np.random.RandomState(2025)

    def add(kind, joints, bl, prm):
        o64, j64 = run(torch.float64, joints, bl, prm, True)
        o32, _ = run(torch.float32, joints, bl, prm, False)
        if not (np.isfinite(o64).all() and np.isfinite(j64).all()):
            print('pose chain case (%s): the reference itself is not finite here -- not a fixture' % kind)
            return
        oris.append(joints); bls.append(bl); prms.append(prm); outs.append(o64); jacs.append(j64); kinds.append(kind)
        floor.append(np.abs(o32 - o64).max() / np.abs(o64).max())
        print('pose chain case %d (%s): |out| max %.3f, fp32 run vs fp64 run %.2e, |jac| max %.3f'
              % (len(oris) - 1, kind, np.abs(o64).max(), floor[-1], np.abs(j64).max()))

    def bone_lengths(j):          # what bench.build_fit_data passes: fitting.bone_lengths_of (kp3D_to_bones on the biomech order)
        kb = j[m2b]
        return np.array([np.linalg.norm(kb[i + 1] - kb[0 if i < 5 else i - 4]) for i in range(20)])

    def leaves_to_prm(leaves, f):  # chain.parameters() order: obj_rot, obj_trans, palm_rot, palm_trans, joint_refine_angle, palm_refine_angle
        return np.concatenate([leaves[4][f], leaves[5][f], leaves[2][f].reshape(-1), leaves[3][f]]).astype(np.float64)

    def perturbed_leaves(n, scale, seed):      # tests/test_whole_step.py::_perturb on a HaloPoseChain of n frames at its initial state
        shapes = [(n, 3, 2), (n, 3), (n, 3, 2), (n, 3), (n, 20), (n, 7)]
        eye62 = np.eye(3)[:, :2]
        init = [np.tile(eye62, (n, 1, 1)), np.zeros((n, 3)), np.tile(eye62, (n, 1, 1)), np.zeros((n, 3)), np.zeros((n, 20)), np.zeros((n, 7))]
        return [a + (scale * torch.randn(sh, generator=torch.Generator().manual_seed(seed + i))).float().numpy()
                for i, (a, sh) in enumerate(zip(init, shapes))]
    # (a) the whole-step tests' own hands: bench.build_fit(seed 40, 1 frame) + _perturb(4e-3, 50); (seed 41, 4 frames) + _perturb(4e-3, 60);
    #     the reproducibility tests' (4e-3, 70) and (1e-2, 20)
    for seed, n, scale, pseed, what in ((40, 1, 4e-3, 50, 'whole-step C3'), (41, 4, 4e-3, 60, 'whole-step C5'), (40, 1, 4e-3, 70, 'reproducibility C3'),
                                        (41, 4, 1e-2, 20, 'reproducibility C5')):
        _, _, j = synth.synth_hand_pose(seed)
        j = j.astype(np.float32)
        from honerf_amd.fitting import bone_lengths_of          # exactly what bench.build_fit_data hands the chain (fp32)
        bl = bone_lengths_of(j[None])[0].numpy()
        lv = perturbed_leaves(n, scale, pseed)
        for f in range(n):
            add('%s hand, frame %d' % (what, f), j.astype(np.float64), bl.astype(np.float64), leaves_to_prm(lv, f))
    # (b) large joint angles and palm motions (an optimisation that has moved far from the prediction)
    for c in range(6):
        _, _, joints = synth.synth_hand_pose(200 + c, center=(0.02 * c - 0.05, 0.03, 0.8 + 0.05 * c), flex=0.2 + 0.15 * c)
        joints = joints.astype(np.float64) + 0.004 * rng2.standard_normal((21, 3))
        bl = bone_lengths(joints) * (1.0 + 0.1 * rng2.standard_normal(20))
        prm = np.zeros(36)
        prm[27:33] = np.eye(3)[:, :2].reshape(-1)
        prm[0:20] = 0.5 * rng2.standard_normal(20)
        prm[20:27] = 0.9 * rng2.standard_normal(7)
        prm[27:33] += 0.6 * rng2.standard_normal(6)
        prm[33:36] = 0.05 * rng2.standard_normal(3)
        add('large angles', joints, bl, prm)
    # (c) near-degenerate geometry: a nearly planar hand (the synthetic hand's own plane, 2e-4 off it), nearly straight fingers
    #     (flex 0.01), very short and very long bones
    for c, (flex, off, bl_scale) in enumerate(((0.3, 2e-4, 1.0), (0.01, 4e-3, 1.0), (0.01, 5e-4, 1.0), (0.35, 4e-3, 0.05), (0.35, 4e-3, 3.0),
                                               (0.6, 1e-3, 0.3))):
        _, _, joints = synth.synth_hand_pose(300 + c, center=(0.0, 0.0, 0.9), flex=flex)
        joints = joints.astype(np.float64) + off * rng2.standard_normal((21, 3))
        bl = bone_lengths(joints) * bl_scale
        prm = np.zeros(36)
        prm[27:33] = np.eye(3)[:, :2].reshape(-1)
        prm[0:20] = 0.1 * rng2.standard_normal(20)
        prm[20:27] = 0.3 * rng2.standard_normal(7)
        prm[27:33] += 0.15 * rng2.standard_normal(6)
        prm[33:36] = 0.01 * rng2.standard_normal(3)
        add('near-degenerate: flex %.2f, %.0e off the plane, bone lengths x %.2f' % (flex, off, bl_scale), joints, bl, prm)
    # (d) the bench's sequence (seed 60, drift 0.002): two of its frames at moderate refinements
    rs = np.random.RandomState(60)
    _, _, j = synth.synth_hand_pose(60)
    rs.standard_normal((2000, 3))
    steps = np.cumsum(rs.standard_normal((8, 1, 3)).astype(np.float32) * 0.002, axis=0)
    for f in (3, 7):
        jj = (j[None] + steps)[f].astype(np.float64)
        prm = np.zeros(36)
        prm[27:33] = np.eye(3)[:, :2].reshape(-1)
        prm[0:20] = 0.05 * rng2.standard_normal(20)
        prm[20:27] = 0.1 * rng2.standard_normal(7)
        add('bench sequence frame %d' % f, jj, bone_lengths_of(jj.astype(np.float32)[None])[0].numpy().astype(np.float64), prm)
    N = len(oris)
    outs = np.stack(outs)
    save('pose_chain', ori_pose=np.stack(oris), bone_len=np.stack(bls), params=np.stack(prms), bt_inv=outs[:, :336].reshape(N, 21, 4, 4),
         joint_3d=outs[:, 336:].reshape(N, 21, 3), jac=np.stack(jacs), ref32_vs_ref64=np.array(floor), kinds=np.array(kinds))


def main():
    only = os.environ.get('HONERF_GOLDEN_ONLY', '')
    if only == 'pose':
        pose_goldens()
        return
    if only == 'loss':
        emb, nets = build_nets()
        emb_b, nets_b = build_nets(use_batch=True)
        loss_goldens(nets, nets_b, torch.Generator().manual_seed(4321))
        return
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

    # ---- loss formulas of the fitting scripts + get_stable_loss_cross (own generator: the fixtures above keep theirs) ----
    loss_goldens(nets, nets_b, torch.Generator().manual_seed(4321))


if __name__ == '__main__':
    main()

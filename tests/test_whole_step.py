"""Whole-step parity of the fitting loops at their real sizes.

ONE `fitting.fit_backward` of the product -- pose leaves -> hn_pose_chain -> rays -> hn_render_dual (importance sampling, far-field
compaction, two streams, tape) -> loss node -> hn_render_dual_bwd -> pose-chain VJP -> the SIX LEAF GRADIENTS -- against an ORACLE
COMPOSITION of the same step on the CPU:

    oracle.pose_chain (float64 chain + exact Jacobian; pinned by pose_chain.npz)
 -> oracle.render: both fields + alpha + two-field compositing at the product's final depths (the sampling is under no_grad in the
    reference, utils/renderer.py:461: depths carry no gradient; they are held to the reference separately, bit-exact indices)
 -> oracle.losses (fitting_single.py:251-283 / fitting_video.py:285-334 / get_stable_loss_cross; pinned by loss_*.npz, stable_loss.npz)
 -> torch autograd.

C3: 196 rays x 192 depths, fit type 12, HaloPoseChain, fixed t_rand, far-field compaction on and off.
C5: one fitting_video window, 4 frames x 40 rays, fit type 1234 with the stable term, anchored at the sequence start.

The oracle runs twice: in fp32 (the reference's arithmetic) and in float64 (the exact value).  Bound per quantity: the north
star's 1e-4 of the tensor's maximum against the fp32 oracle, or -- where the fp32 oracle is itself further than that from the
float64 value (measured here, recorded in the parity report) -- at least as close to float64 as the fp32 oracle is
(helpers.assert_parity).  The contact / penetration terms select samples by thresholds (|s_h| + |s_o| < 1e-2; s < 0): a sample
within rounding of a threshold may be selected on one side only, which moves a mean over ~10^3 selected samples by ~1e-3 of
itself; the number of such samples (fp32 vs float64 oracle) is recorded with the comparison, and the three terms that depend on
them are held to 5e-3 only when it is non-zero.
"""
import numpy as np
import pytest
import torch

from helpers import record, rel_err

pytestmark = pytest.mark.gpu


class _PoseChainOracle(torch.autograd.Function):
    """oracle.pose_chain as an autograd node: values and the exact Jacobian in float64."""

    @staticmethod
    def forward(ctx, params, ori_pose, bone_len):
        from oracle.pose_chain import pose_chain
        bt, j3, jac = pose_chain(ori_pose.numpy(), bone_len.numpy(), params.detach().double().numpy())
        ctx.jac = torch.from_numpy(jac)
        ctx.dtype = params.dtype
        return torch.from_numpy(bt).to(params.dtype), torch.from_numpy(j3).to(params.dtype)

    @staticmethod
    def backward(ctx, g_bt, g_j3):
        F = ctx.jac.shape[0]
        g = torch.cat([g_bt.reshape(F, 336), g_j3.reshape(F, 63)], dim=1).double()
        return torch.einsum('fo,foi->fi', g, ctx.jac).to(ctx.dtype), None, None


def _oracle_fields(nets, dtype):
    from oracle.nets import Field
    sd = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    hand, obj = Field('hand', sd(nets[0]), sd(nets[2]), 0.3), Field('obj', sd(nets[3]), sd(nets[5]), 0.3)
    if dtype == torch.float64:
        for f in (hand, obj):
            f.sdf = [(W.double(), b.double()) for W, b in f.sdf]
            f.color = [(W.double(), b.double()) for W, b in f.color]
            f.variance = f.variance.double()
    return hand, obj


def _oracle_step(nets, chain, index, view, z, fit_type, dtype, video, ends=(False, False), obj_verts_stable=None, chunk_rays=49):
    """The oracle composition of one step -> (terms, [six leaf gradients], diagnostics).  z: the product's final depths [N,S]."""
    from oracle import losses as ol
    from oracle import render as orr
    hand, obj = _oracle_fields(nets, dtype)
    c = lambda x: x.detach().cpu().to(dtype) if x.is_floating_point() else x.detach().cpu()
    rows = slice(None) if index is None else torch.as_tensor(index)
    leaves = [c(p).clone().requires_grad_(True) for p in chain.parameters()]          # obj_rot, obj_trans, palm_rot, palm_trans, joint_refine, palm_refine
    obj_rot, obj_trans, palm_rot, palm_trans, jra, pra = [x[rows] for x in leaves]
    Fr = jra.shape[0]
    joints0, bone_len = chain.joints0[rows].detach().cpu().double(), chain.bone_len[rows].detach().cpu().double()
    Ro_pred, To_pred, T_pose = c(chain.Ro_pred[rows]), c(chain.To_pred[rows]), c(chain.T_pose_21[rows])
    verts = c(chain.obj_verts)
    params = torch.cat([jra, pra, palm_rot.reshape(Fr, 6), palm_trans], dim=1)
    bt_inv, joint_3d = _PoseChainOracle.apply(params, joints0, bone_len)                # fitting_single.py:206-226
    obj_r = ol.rot6d_to_matrix(obj_rot) @ Ro_pred                                        # :227-231
    obj_t = To_pred + obj_trans
    pred_v = (obj_r.unsqueeze(1) @ verts[None, :, :, None])[..., 0] + obj_t.unsqueeze(1)
    comp_v = (Ro_pred.unsqueeze(1) @ verts[None, :, :, None])[..., 0] + To_pred.unsqueeze(1)
    # the render's pose inputs as leaves of their own: the render is differentiated chunk by chunk, the chain once
    Ro_in = (torch.inverse(obj_r) if video else obj_r[0].T).detach().clone().requires_grad_(True)   # fitting_video.py:284 / fitting_single.py:250
    To_in = (obj_t if video else obj_t[0]).detach().clone().requires_grad_(True)
    bt_in = (bt_inv if video else bt_inv[0]).detach().clone().requires_grad_(True)
    cam = {k: c(v) for k, v in view['cam'].items()}
    n_cams = cam['R'].shape[0]
    xy = c(view['xy'])
    P = xy.shape[0] // n_cams
    rays = [orr.rays_from_xy(xy[k * P:(k + 1) * P], cam['R'][k], cam['T'][k], cam['focal'][k], cam['principal'][k]) for k in range(n_cams)]
    S = z.shape[-1]
    zc = z.detach().cpu().to(dtype).reshape(n_cams, P, S)
    near, far, n_samples = 0.4, 1.5, 64
    sample_dist = (far - near) / n_samples

    def render_chunk(k, a, b):
        """rays a:b of camera / frame k -> (color [n,3], wsum [n,1], sdf_h [n*S,1], sdf_o [n*S,1])."""
        o, d = rays[k][0][a:b].to(dtype), rays[k][1][a:b].to(dtype)
        zz = zc[k, a:b]
        if video:
            o, d, zz = o[None], d[None], zz[None]
            bt_k, tp_k, Ro_k, To_k = bt_in[k:k + 1], T_pose[k:k + 1], Ro_in[k:k + 1], To_in[k:k + 1]
        else:
            bt_k, tp_k, Ro_k, To_k = bt_in, T_pose[0], Ro_in, To_in
        oo, do = orr.obj_local(o, d, Ro_k, To_k, repeat=True)
        a_h, c_h, s_h, _, _ = orr._alpha_sample_color(hand, o, d, zz, sample_dist, bt_k, tp_k, video)
        a_o, c_o, s_o, _, _ = orr._alpha_sample_color(obj, oo, do, zz, sample_dist, bt_k, tp_k, video)
        color, wsum, _, _ = orr.composite_dual(a_h, c_h, a_o, c_o)
        return color.reshape(-1, 3), wsum.reshape(-1, 1), s_h, s_o

    chunks = [(k, a, min(a + chunk_rays, P)) for k in range(n_cams) for a in range(0, P, chunk_rays)]
    # pass 1: the render outputs (graphs dropped chunk by chunk)
    outs = [tuple(x.detach() for x in render_chunk(*ch)) for ch in chunks]
    full = [torch.cat([o[i] for o in outs]).requires_grad_(True) for i in range(4)]
    color, wsum, sdf_h, sdf_o = full
    ro = {'color_fine': color.reshape(n_cams, P, 3) if video else color, 'weight_sum': wsum.reshape(n_cams, P, 1) if video else wsum,
          'sdf_hand': sdf_h, 'sdf_obj': sdf_o}
    true_rgb, true_mask = c(view['true_rgb']), c(view['true_mask'])
    if video:
        stable = None
        if obj_verts_stable is not None:
            stable = ol.stable_loss_cross(lambda p, b, tp: hand.sdf_only(p, b, tp), c(obj_verts_stable), bt_inv, T_pose, obj_r, obj_t)
            stable = stable if isinstance(stable, torch.Tensor) else torch.zeros((), dtype=dtype)
        terms = ol.video_step_loss(ro, true_rgb, true_mask, joint_3d, joints0.to(dtype), pred_v, comp_v, [0 if ends[0] else 1, 0, 0, 9 if ends[1] else 0], 10,
                                   ends[0] or ends[1], stable=stable)
    else:
        terms = ol.single_step_loss(ro, true_rgb, true_mask, joint_3d, joints0[0].to(dtype), ol.pose_loss_single(comp_v[0], pred_v[0]), fit_type)
    # the loss differentiated w.r.t. the render outputs (leaves `full`) and, through the chain, directly w.r.t. the six pose leaves
    terms['loss'].backward(retain_graph=True)       # (the chain's graph is walked once more below)
    # pass 2: the render's pose inputs, chunk by chunk
    off_r, off_s = 0, 0
    for ch in chunks:
        n_r = ch[2] - ch[1]
        got = render_chunk(*ch)
        gs = [color.grad[off_r:off_r + n_r], wsum.grad[off_r:off_r + n_r],
              sdf_h.grad[off_s:off_s + n_r * S] if sdf_h.grad is not None else torch.zeros_like(got[2]),
              sdf_o.grad[off_s:off_s + n_r * S] if sdf_o.grad is not None else torch.zeros_like(got[3])]
        torch.autograd.backward(list(got), gs)
        off_r, off_s = off_r + n_r, off_s + n_r * S
    # ... and on through the chain into the leaves
    heads = [bt_inv if video else bt_inv[0], torch.inverse(obj_r) if video else obj_r[0].T, obj_t if video else obj_t[0]]
    torch.autograd.backward(heads, [bt_in.grad, Ro_in.grad, To_in.grad])
    grads = [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves]
    sa = (sdf_h.detach().abs() + sdf_o.detach().abs())[:, 0]
    diag = {'sdf_hand': sdf_h.detach()[:, 0], 'sdf_obj': sdf_o.detach()[:, 0], 'contact_n': int((sa < 1e-2).sum()),
            'penet_n': int(((sdf_h.detach() < 0) & (sdf_o.detach() < 0)).sum())}
    return {k: v.detach() for k, v in terms.items() if isinstance(v, torch.Tensor)}, [g.detach() for g in grads], diag


def _perturb(chain, scale, seed):
    with torch.no_grad():                                                   # away from the identity start: every leaf's gradient is generic
        for i, p in enumerate(chain.parameters()):
            p.add_(scale * torch.randn(p.shape, generator=torch.Generator().manual_seed(seed + i)).to(p.device))


# Observed on MI355X (profiles/r04/parity_report.json): loss terms 0 ... 9e-7; leaf gradients 4.7e-6 ... 6.8e-4 from the fp32 oracle
# where the fp32 oracle is itself 5.5e-6 ... 1.7e-2 from float64 and the product 5.2e-6 ... 1.7e-2 (never further than 1.17x the fp32
# oracle's distance; closer than it on 7 of the 18 gradients).  GRAD_CAP: 4x the largest product-vs-fp32-oracle error.
GRAD_CAP = 2.8e-3
LEAVES = ('obj_rot_refine', 'obj_trans_refine', 'palm_rot_refine', 'palm_trans_refine', 'joint_refine_angle', 'palm_refine_angle')
TERM_KEYS_SINGLE = ('loss', 'color', 'mask', 'contact', 'penetration', 'joint', 'obj_verts')


def _compare(tag, got_terms, got_grads, ref32, ref64, keys, threshold_flips, grad_cap):
    t32, g32, _ = ref32
    t64, g64, _ = ref64
    for k in keys:
        a, b, e = float(got_terms[k]), float(t32[k]), float(t64[k])
        scale = max(abs(b), 1e-12)
        err, floor = abs(a - b) / scale, abs(b - e) / scale
        # contact / penetration (and with them the loss): means over threshold-selected samples, see the module docstring
        bound = max(1e-4, 4.0 * floor) if (threshold_flips == 0 or k in ('color', 'mask', 'joint', 'obj_verts', 'smooth', 'stable')) else 5e-3
        record('%s term %s' % (tag, k), err, bound, ref32_vs_fp64=floor, value=b, threshold_flips=threshold_flips)
        assert err <= bound, '%s %s: %.6g vs oracle %.6g (rel %.2e > %.1e)' % (tag, k, a, b, err, bound)
    for name, a, b, e in zip(LEAVES, got_grads, g32, g64):
        a = a.cpu().double().numpy()
        e_hr, e_ref, e_hip = rel_err(a, b.double().numpy()), rel_err(b.double().numpy(), e.numpy()), rel_err(a, e.numpy())
        record('%s d loss / d %s' % (tag, name), e_hr, grad_cap, kind='rel, conditioning-aware', ref32_vs_fp64=e_ref, hip_vs_fp64=e_hip)
        # north star, or (where the fp32 oracle itself is further than that from float64) as close to float64 as the fp32 oracle is.
        # Where rounding flips a sample's membership of the contact / penetration sets between the fp32 and the float64 oracle
        # (threshold_flips > 0: one of ~10^2 .. 10^3 selected samples joining or leaving a mean moves that term's gradient by
        # 10^-3 .. 10^-2 of itself) the fp32 oracle is not a reference to within grad_cap either: float64 decides alone.
        # The same where the fp32 oracle is further than grad_cap from float64 WITHOUT a flip (e_ref > grad_cap: the C5 'surface' scene's
        # d / d palm_trans, a sum over all samples with heavy cancellation, 3.0e-2): no result can be within grad_cap of such a
        # reference and within e_ref of float64 at once (triangle inequality); the product there is 7e-3 from float64.
        ok = e_hr <= 1e-4 or ((e_hr <= grad_cap or threshold_flips > 0 or e_ref > grad_cap) and e_hip <= 1.5 * e_ref + 1e-5)
        assert ok, '%s d/d %s: product vs fp32 oracle %.3e, fp32 oracle vs float64 %.3e, product vs float64 %.3e' % (tag, name, e_hr, e_ref, e_hip)


# Two synthetic scenes.  'inside': the object's centre 2 cm from joint 9 -- its sphere-like field (radius ~0.4 in its own frame)
# contains the whole hand: thousands of samples in the PENETRATION set, none in the contact set.  'surface': the centre 0.30 m above
# the hand, so that the object's zero level crosses the fingers -- hundreds of samples with |s_h| + |s_o| < 1e-2 (the CONTACT branch of
# fitting_single.py:268-275 / fitting_video.py:293-300) and a penetration set beside them (tools/contact_scene_probe.py: 1 440 / 2 485
# samples on view 0 of the C3 step).
SCENES = {'inside': (0.02, 0.0, 0.01), 'surface': (0.0, 0.30, 0.0)}


@pytest.mark.parametrize('scene', ['inside', 'surface'])
def test_fitting_single_step_matches_the_oracle_composition(scene):
    """C3 / C4: one fit_backward of fitting_single at 196 rays x 192 depths, fit type 12, the reference's six-leaf chain, fixed
    t_rand, with and without the far-field compaction."""
    import bench
    from honerf_amd import fitting as F
    dev = torch.device('cuda')
    ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True, obj_offset=SCENES[scene])
    _perturb(chain, 4e-3, 50)
    tr = torch.rand(bench.FIT_RAYS, 1, generator=torch.Generator().manual_seed(7)).to(dev)
    view = views[0]
    runs = {}
    for compact in (True, False):
        ren.compact_far_field = compact
        terms = F.fit_backward(ren, view, chain, bench.NEAR, bench.FAR, '12', t_rand=tr)
        torch.cuda.synchronize()
        runs[compact] = ({k: v.detach().cpu() for k, v in terms.items()}, [p.grad.detach().clone() for p in chain.parameters()],
                         ren.last_z_vals.detach().clone(), ren._tape.buf)
    assert torch.equal(runs[True][2], runs[False][2]), 'the far-field skip moved a depth'
    z = runs[True][2]
    assert z.shape == (bench.FIT_RAYS, bench.FIT_N + 2 * bench.FIT_IMP)
    ref32 = _oracle_step(nets, chain, None, view, z, '12', torch.float32, video=False)
    ref64 = _oracle_step(nets, chain, None, view, z, '12', torch.float64, video=False)
    record('C3 step (%s scene): samples in the contact set' % scene, ref32[2]['contact_n'], float('inf'), kind='value')
    record('C3 step (%s scene): samples in the penetration set' % scene, ref32[2]['penet_n'], float('inf'), kind='value')
    assert ref32[2]['penet_n'] + ref32[2]['contact_n'] > 0, 'the interaction terms are not exercised by this scene'
    if scene == 'surface':      # the contact branch's gradient flows through the render's adjoint end to end
        assert ref32[2]['contact_n'] >= 200 and float(ref32[0]['contact']) > 0.0, ref32[2]
        assert ref32[2]['penet_n'] >= 200, ref32[2]
    # samples whose threshold membership the fp32 and the float64 oracle disagree on (what rounding can flip)
    sel = lambda d: ((d['sdf_hand'].abs() + d['sdf_obj'].abs()) < 1e-2, (d['sdf_hand'] < 0) & (d['sdf_obj'] < 0))
    flips = int(sum((a != b).sum() for a, b in zip(sel(ref32[2]), sel(ref64[2]))))
    for compact in (True, False):
        _compare('C3 step (%s scene, compaction %s)' % (scene, 'on' if compact else 'off'), runs[compact][0], runs[compact][1], ref32, ref64, TERM_KEYS_SINGLE,
                 flips, grad_cap=GRAD_CAP)


@pytest.mark.parametrize('ends,scene', [((True, False), 'inside'), ((False, True), 'inside'), ((True, False), 'surface')])
def test_fitting_video_window_step_matches_the_oracle_composition(ends, scene):
    """C5: one fit_backward of a fitting_video window -- 4 frames x 40 rays, fit type 1234 (stable term on the object's vertices),
    anchored at the sequence start / at its end, batched renderer with the reference's SDF-row quirk B-1 in the sampling."""
    import bench
    from honerf_amd import fitting as F
    dev = torch.device('cuda')
    ren, nets, chain, views, verts = bench.build_fit(dev, 41, 4, bench.VID_RAYS, 'f16x3', halo=True, obj_offset=SCENES[scene])
    assert ren.strict_reference and ren.batched
    _perturb(chain, 4e-3, 60)
    tr = torch.rand(4 * bench.VID_RAYS, 1, generator=torch.Generator().manual_seed(8)).to(dev)
    view = views[0]
    ov = verts[:, :400].contiguous()
    terms = F.fit_backward(ren, view, chain, bench.NEAR, bench.FAR, '1234', index=[0, 1, 2, 3], smooth_ends=ends, obj_verts_for_stable=ov,
                           t_rand=tr)
    torch.cuda.synchronize()
    got_terms = {k: v.detach().cpu() for k, v in terms.items()}
    got_grads = [p.grad.detach().clone() for p in chain.parameters()]
    z = ren.last_z_vals.detach().reshape(4 * bench.VID_RAYS, -1)
    kw = dict(video=True, ends=ends, obj_verts_stable=ov, chunk_rays=bench.VID_RAYS)
    ref32 = _oracle_step(nets, chain, [0, 1, 2, 3], view, z, '1234', torch.float32, **kw)
    ref64 = _oracle_step(nets, chain, [0, 1, 2, 3], view, z, '1234', torch.float64, **kw)
    sel = lambda d: ((d['sdf_hand'].abs() + d['sdf_obj'].abs()) < 1e-2, (d['sdf_hand'] < 0) & (d['sdf_obj'] < 0))
    flips = int(sum((a != b).sum() for a, b in zip(sel(ref32[2]), sel(ref64[2]))))
    keys = ['loss', 'color', 'mask', 'contact', 'penetration', 'joint', 'obj_verts', 'smooth']
    if float(ref32[0].get('stable', torch.zeros(()))) != 0.0:
        keys.append('stable')
    record('C5 window step (%s scene): stable term of the oracle' % scene, float(ref32[0].get('stable', torch.zeros(()))), float('inf'), kind='value')
    record('C5 window step (%s scene): samples in the contact set' % scene, ref32[2]['contact_n'], float('inf'), kind='value')
    if scene == 'surface':
        assert ref32[2]['contact_n'] >= 50 and float(ref32[0]['contact']) > 0.0, ref32[2]
    _compare('C5 window step (%s scene, anchor %s)' % (scene, 'first' if ends[0] else 'last'), got_terms, got_grads, ref32, ref64, keys, flips, grad_cap=GRAD_CAP)

"""Pins the oracle (CPU restatement) to vectors produced by the reference
itself (tests/golden/make_golden.py).  Runs on CPU, no GPU needed.

Tolerance: the oracle and the reference run the same torch CPU kernels in
almost the same order, so agreement is ~1e-6; the bound asserted is 2e-5
relative (integer indices: exact)."""
import numpy as np
import torch

from helpers import assert_close, oracle_fields, t
from oracle import nets as on
from oracle import render as orr

RT = 2e-5


def test_embed(golden):
    g = golden('embed')
    x = t(g['x'])
    for L in (10, 4, 7):
        assert_close(on.embed(x, L), g['L%d' % L], 1e-6, 'embed L=%d' % L)


def test_bone_coords(golden):
    g = golden('bone_coords')
    v, r, h = on.bone_coords(t(g['pts']), t(g['bt_inv']), t(g['T_pose']))
    assert_close(v, g['v'], 1e-6, 'v')
    assert_close(r, g['r'], 1e-6, 'r')
    assert_close(h, g['h'], 1e-6, 'h')
    v, r, h = on.bone_coords(t(g['pts_b']), t(g['bt_inv_b']), t(g['T_pose_b']))
    assert_close(v, g['v_b'], 1e-6, 'v batch')
    assert_close(r, g['r_b'], 1e-6, 'r batch')
    assert_close(h, g['h_b'], 1e-6, 'h batch')


def test_field_obj(golden):
    g = golden('field_obj')
    _, obj = oracle_fields()
    pts, dirs = t(g['pts']), t(g['dirs'])
    out = on.obj_sdf_forward(obj.sdf, pts)
    assert_close(out, g['out'], RT, 'obj sdf out')
    sdf, grad, rgb = obj.evaluate(pts, dirs)
    assert_close(sdf, g['out'][:, :1], RT, 'obj sdf')
    assert_close(grad, g['grad'], RT, 'obj grad')
    assert_close(rgb, g['rgb'], RT, 'obj rgb')


def test_field_hand(golden):
    g = golden('field_hand')
    hand, _ = oracle_fields()
    pts, dirs, bt, Tp = t(g['pts']), t(g['dirs']), t(g['bt_inv']), t(g['T_pose'])
    out, feat = on.hand_sdf_forward(hand.sdf, pts, bt, Tp)
    assert_close(out, g['out'], RT, 'hand sdf out')
    assert_close(feat[:8], g['feat'], 1e-6, 'hand features')
    sdf, grad, rgb = hand.evaluate(pts, dirs, bt, Tp)
    assert_close(grad, g['grad'], RT, 'hand grad')
    assert_close(rgb, g['rgb'], RT, 'hand rgb')


def test_upsample_and_merge(golden):
    g = golden('upsample')
    z, sdf = t(g['z']), t(g['sdf'])
    for i in range(4):
        z_new, inds = orr.up_sample(z, sdf, 16, 64 * 2 ** i)
        assert np.array_equal(inds.numpy(), g['inds%d' % i]), 'inds step %d' % i
        assert_close(z_new, g['znew%d' % i], 1e-6, 'z_new step %d' % i)
        z, sdf, index = orr.merge_z(z, z_new, sdf, t(g['sdfnew%d' % i]))
        assert np.array_equal(index.numpy(), g['index%d' % i]), 'sort index step %d' % i
        assert np.array_equal(z.numpy(), g['zmerged%d' % i])
        assert np.array_equal(sdf.numpy(), g['sdfmerged%d' % i])


def _check_single(g, field, **kw):
    res = orr.render_single(field, t(g['rays_o']), t(g['rays_d']), float(g['near']), float(g['far']),
                            t(g['t_rand']), int(g['n_samples']), int(g['n_importance']), 4, **kw)
    for k in ('color_fine', 's_val', 'cdf_fine', 'weight_sum', 'weight_max', 'gradient_error'):
        assert_close(res[k], g[k], RT, k)
    n_steps = len(res['inds'])
    for i in range(n_steps):
        assert np.array_equal(res['inds'][i].numpy(), g['inds%d' % i]), 'inds %d' % i
        assert np.array_equal(res['index'][i].numpy(), g['index%d' % i]), 'index %d' % i


def test_render_obj(golden):
    _, obj = oracle_fields()
    for tag in ('obj_64_64', 'obj_32_0'):
        g = golden('render_' + tag)
        _check_single(g, obj, Ro=t(g['Ro']), To=t(g['To']))


def test_render_hand(golden):
    hand, _ = oracle_fields()
    for tag in ('hand_64_64', 'hand_64_0'):
        g = golden('render_' + tag)
        _check_single(g, hand, bt_inv=t(g['bt_inv']), T_pose=t(g['T_pose']))


def test_render_dual_forward_backward(golden):
    g = golden('render_dual')
    hand, obj = oracle_fields()
    Ro = t(g['Ro']).requires_grad_(True)
    To = t(g['To']).requires_grad_(True)
    bt = t(g['bt_inv']).requires_grad_(True)
    ro = t(g['rays_o']).requires_grad_(True)
    rd = t(g['rays_d']).requires_grad_(True)
    res = orr.render_dual(hand, obj, ro, rd, float(g['near']), float(g['far']), t(g['t_rand']), 64, 64, 4,
                          bt, t(g['T_pose']), Ro, To)
    for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj', 'gradient_error_hand', 'gradient_error_obj',
              'gradient_hand', 'gradient_obj'):
        # with every input requiring grad torch takes other matmul-backward paths than with
        # plain tensors (bitwise equal without requires_grad): allow the north-star 1e-4
        assert_close(res[k], g[k], 1e-4, k)
    # sample indices: bit-exact on plain tensors (same torch kernels as the reference run)
    plain = orr.render_dual(hand, obj, t(g['rays_o']), t(g['rays_d']), float(g['near']), float(g['far']),
                            t(g['t_rand']), 64, 64, 4, t(g['bt_inv']), t(g['T_pose']), t(g['Ro']), t(g['To']))
    inds = torch.stack([x for pair in zip(plain['inds_hand'], plain['inds_obj']) for x in pair])
    assert np.array_equal(inds.numpy(), g['inds'])
    for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj', 'gradient_hand', 'gradient_obj'):
        assert_close(plain[k], g[k], RT, k + ' (plain)')
    loss = ((res['color_fine'] * t(g['w_color'])).sum() + (res['weight_sum'] * t(g['w_wsum'])).sum()
            + (res['sdf_hand'] * t(g['w_sdf_hand'])).sum() + (res['sdf_obj'] * t(g['w_sdf_obj'])).sum())
    assert_close(loss, g['loss'], RT, 'loss')
    grads = torch.autograd.grad(loss, [Ro, To, bt, ro, rd])
    for name, gr in zip(('g_Ro', 'g_To', 'g_bt_inv', 'g_rays_o', 'g_rays_d'), grads):
        assert_close(gr, g[name], 1e-4, name)


def test_render_dual_batch(golden):
    g = golden('render_dual_batch')
    hand, obj = oracle_fields()
    res = orr.render_dual(hand, obj, t(g['rays_o']), t(g['rays_d']), float(g['near']), float(g['far']),
                          t(g['t_rand']), 64, 64, 4, t(g['bt_inv']), t(g['T_pose']), t(g['Ro']), t(g['To']),
                          batch_quirk=True)
    for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj', 'gradient_error_hand', 'gradient_error_obj',
              'gradient_hand', 'gradient_obj'):
        assert_close(res[k], g[k], RT, k)


# ---- the loss half of an oracle fitting step (oracle/losses.py) against the reference's own statements ------------------
def _leaf(a):
    return t(a).clone().requires_grad_(True)


def test_single_step_loss_oracle(golden):
    """oracle.losses.single_step_loss against loss_single.npz (fitting_single.py:251-288 executed on synthetic render
    outputs): every term and the gradient w.r.t. every tensor the block consumes, both fit types."""
    from oracle import losses as ol
    g = golden('loss_single')
    for ft in ('1', '12'):
        ro = {k: _leaf(g['in_' + k]) for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj')}
        j3 = _leaf(g['joint_3d'])
        terms = ol.single_step_loss(ro, t(g['true_rgb']), t(g['true_mask']), j3, t(g['joint3d_pred']), torch.tensor(float(g['obj_verts_loss'])), ft)
        pre = 's%s_' % ft
        for key, name in (('loss', 'loss'), ('color', 'color'), ('mask', 'mask'), ('joint', 'joint')):
            assert_close(terms[key], g[pre + name], 2e-6, 'oracle ' + pre + name)
        if ft == '12':
            assert_close(terms['contact'], g['s12_contact'], 2e-6, 'oracle contact')
            assert_close(terms['penetration'], g['s12_penet'], 2e-6, 'oracle penetration')
        leaves = [ro['color_fine'], ro['weight_sum'], ro['sdf_hand'], ro['sdf_obj'], j3]
        grads = torch.autograd.grad(terms['loss'], leaves, allow_unused=True)
        for name, gr, like in zip(('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj', 'joint_3d'), grads, leaves):
            gr = torch.zeros_like(like) if gr is None else gr
            ref = g[pre + 'g_' + name]
            if np.abs(ref).max() == 0:
                assert float(gr.abs().max()) == 0.0, name
            else:
                assert_close(gr, ref, 2e-6, 'oracle ' + pre + 'g_' + name)


def test_video_step_loss_oracle(golden):
    """oracle.losses.video_step_loss against loss_video.npz (fitting_video.py:285-339): a middle window, both sequence ends
    and the very first step."""
    from oracle import losses as ol
    g = golden('loss_video')
    for tag in ('mid', 'head', 'tail', 'first'):
        ro = {k: _leaf(g['in_' + k]) for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj')}
        j3, pv = _leaf(g['joint_3d']), _leaf(g['pred_obj_v_w'])
        terms = ol.video_step_loss(ro, t(g['true_rgb']), t(g['true_mask']), j3, t(g['joint3d_pred']), pv, t(g['compare_obj_v_w']),
                                   g[tag + '_index'], 10, not bool(g[tag + '_first']), stable=t(g['stable']).reshape(()))
        for key, name in (('loss', 'loss'), ('smooth', 'smooth'), ('color', 'color'), ('mask', 'mask'), ('contact', 'contact'),
                          ('penetration', 'penet')):
            assert_close(terms[key], g[tag + '_' + name], 3e-6, 'oracle %s_%s' % (tag, name))
        grads = torch.autograd.grad(terms['loss'], [ro['color_fine'], ro['weight_sum'], ro['sdf_hand'], ro['sdf_obj'], j3, pv])
        for name, gr in zip(('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj', 'joint_3d', 'pred_obj_v_w'), grads):
            assert_close(gr, g['%s_g_%s' % (tag, name)], 3e-6, 'oracle %s_g_%s' % (tag, name))


def test_stable_loss_oracle(golden):
    """oracle.losses.stable_loss_cross over the oracle hand field against stable_loss.npz (the reference's
    get_stable_loss_cross on its batched renderer): value and gradients w.r.t. bt_inv, the object rotation and translation."""
    from oracle import losses as ol
    g = golden('stable_loss')
    hand, _ = oracle_fields()
    bt, R, T = _leaf(g['bt_inv']), _leaf(g['obj_r']), _leaf(g['obj_t'])
    sdf_fn = lambda p, b, tp: hand.sdf_only(p, b, tp)
    with torch.no_grad():
        pw = (R.unsqueeze(1) @ t(g['obj_verts'])[:, ::10, :].unsqueeze(-1))[..., 0] + T.unsqueeze(1)
        assert_close(sdf_fn(pw, bt, t(g['T_pose'])).reshape(4, -1), g['hand_sdf'], RT, 'oracle hand sdf on the object vertices')
    loss = ol.stable_loss_cross(sdf_fn, t(g['obj_verts']), bt, t(g['T_pose']), R, T)
    assert_close(loss, g['stable'], RT, 'oracle stable loss')
    gb, gR, gT = torch.autograd.grad(loss, [bt, R, T])
    assert_close(gb, g['g_bt_inv'], 5e-5, 'oracle stable d/d bt_inv')
    assert_close(gR, g['g_obj_r'], 5e-5, 'oracle stable d/d obj_r')
    assert_close(gT, g['g_obj_t'], 5e-5, 'oracle stable d/d obj_t')

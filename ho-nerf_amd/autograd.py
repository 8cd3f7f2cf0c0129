"""Differentiable two-field render: `torch.autograd.Function` over the C ABI.

The reference lets autograd differentiate `NeuSRenderer_fitting.render` w.r.t. the pose-dependent inputs
(`rays_o, rays_d, Ro, To, bt_inv`, formally `T_pose_21`): fitting_single.py:289-291, fitting_video.py:340-342.  The
sampling positions carry no gradient (utils/renderer.py:461 `no_grad`), so the backward pass only runs through the
final `get_alpha_sample_color` of both fields and the compositing (utils/renderer.py:360-422, 512-524).  Here that
backward is the launch sequence

    composite2_bwd -> alpha_bwd (x2) -> field_eval_bwd (x2) -> sample_points_bwd (x2) -> obj_local_bwd

on the depths the forward produced -- one C-ABI call, hn_render_dual_bwd, which runs the hand and the object branch
side by side on two streams; the per-field forward values it needs (rgb, alpha) are the ones the forward pass left
in the render workspace.  Nothing here computes on the host; torch only owns the buffers.
"""
import ctypes
import threading
import weakref

import torch

from . import lib as _lib


def _empty(*shape, dev):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


class _AuxHolder:
    """rgb / alpha of both fields of one differentiable render: views into the renderer's tape (owned=False) or copies."""

    def __init__(self, tensors, owned):
        self.t, self.owned = tensors, owned

    def own(self):
        if not self.owned:
            self.t = tuple(x.clone() for x in self.t)
            self.owned = True


class DualRenderFn(torch.autograd.Function):
    """(rays_o [F,P,3], rays_d [F,P,3], bt_inv [F,21,4,4], T_pose [F,21,3], Ro [F,3,3], To [F,3]) ->
    (color [N,3], weight_sum [N,1], sdf_hand [N*S,1], sdf_obj [N*S,1], grad_hand [N*S,3], grad_obj [N*S,3], gerr [2])."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, bt_inv, T_pose, Ro, To, renderer, near, far, t_rand):
        prev = getattr(renderer, '_pending_aux', None)
        prev = prev() if prev is not None else None
        if prev is not None:
            prev.own()        # an earlier render's backward pass has not run yet: its arrays leave the tape before it is overwritten
        o = renderer._render_raw(rays_o.detach(), rays_d.detach(), near, far, bt_inv.detach(), T_pose.detach(), Ro.detach(),
                                 To.detach(), t_rand, keep_tape=True)
        # The tape buffer belongs to the renderer and is overwritten by its next differentiable render: a backward pass
        # uses it only if it is still this render's (the usual forward -> backward order of a fitting step).
        renderer._tape_serial = getattr(renderer, '_tape_serial', 0) + 1
        ctx.tape, ctx.tape_serial = o['tape'], renderer._tape_serial
        ctx.renderer, ctx.near, ctx.far = renderer, float(near), float(far)
        hand, obj = renderer.fields()
        # The backward pass runs on THESE packed fields: the tape's layout (hand samples + the far-field stand-in, the compaction
        # record behind it) follows the hand field's compaction setting at the time of this render, and a renderer whose
        # `compact_far_field` / precision / parameters change before the backward pass builds new fields.
        ctx.fields, ctx.compaction = (hand, obj), bool(getattr(hand, 'compaction', False))
        lib = _lib.load()
        N = rays_o.shape[0] * rays_o.shape[1]
        S = o['z_vals'].shape[-1]
        n = N * S
        aux_off = lib.hn_render_dual_tape_aux_offset(hand.handle, obj.handle, N, S) if o['tape'] is not None else 0
        if aux_off:
            # rgb / alpha of both fields were written into the tape: views, no copies.  They stay valid as long as the tape
            # does; the renderer's next differentiable render takes ownership away first (_AuxHolder.own)
            a = o['tape'][aux_off:aux_off + 32 * n].view(torch.float32)
            ctx.aux = _AuxHolder((a[:3 * n].view(n, 3), a[3 * n:6 * n].view(n, 3), a[6 * n:7 * n], a[7 * n:8 * n]), owned=False)
        else:
            # no tape (fp32 fields): what the final evaluation left in the render workspace, copied out now -- the
            # workspace is re-used by the next render
            offs = (ctypes.c_size_t * 4)()
            _lib.check(lib.hn_render_dual_aux_offsets(hand.handle, obj.handle, N, renderer.n_samples, renderer.n_importance,
                                                      renderer.up_sample_steps, offs),
                       'hn_render_dual_aux_offsets')
            ws = renderer._ws.buf
            view = lambda off, cnt: ws[off:off + 4 * cnt].view(torch.float32).clone()
            ctx.aux = _AuxHolder((view(offs[0], 3 * n).reshape(n, 3), view(offs[1], 3 * n).reshape(n, 3), view(offs[2], n), view(offs[3], n)),
                                 owned=True)
        renderer._pending_aux = weakref.ref(ctx.aux)
        ctx.save_for_backward(rays_o.detach(), rays_d.detach(), bt_inv.detach(), T_pose.detach(), Ro.detach(), To.detach(),
                              o['z_vals'], o['sdf_hand'], o['sdf_obj'], o['grad_hand'], o['grad_obj'])
        # outputs the loss does not use arrive as None in backward, not as zero tensors (three fills and two adds of zeros per step)
        ctx.set_materialize_grads(False)
        ctx.want_rays = bool(rays_o.requires_grad or rays_d.requires_grad)
        return o['color'], o['weight_sum'], o['sdf_hand'], o['sdf_obj'], o['grad_hand'], o['grad_obj'], o['gerr']

    @staticmethod
    def backward(ctx, g_color, g_wsum, g_sdf_h, g_sdf_o, g_grad_h, g_grad_o, g_gerr):
        L = _lib
        lib = L.load()
        ren = ctx.renderer
        rays_o, rays_d, bt, tp, Ro, To, z, sdf_h, sdf_o, grad_h, grad_o = ctx.saved_tensors
        rgb_h, rgb_o, alpha_h, alpha_o = ctx.aux.t
        hand, obj = ctx.fields
        if bool(getattr(hand, 'compaction', False)) != ctx.compaction:
            raise RuntimeError('the hand field\'s far-field compaction was switched between a render and its backward pass: '
                               'the tape of that render cannot be read with the other setting')
        F, P = rays_o.shape[0], rays_o.shape[1]
        N, S = F * P, z.shape[-1]
        n = N * S
        dev = rays_o.device
        st = L.stream_ptr()
        ro, rd = L.f32(rays_o).reshape(N, 3), L.f32(rays_d).reshape(N, 3)
        bt, tp = L.f32(bt).reshape(F, 21, 4, 4), L.f32(tp).reshape(-1, 21, 3)
        if tp.shape[0] != F:
            tp = tp.expand(F, 21, 3).contiguous()
        Ro, To = L.f32(Ro).reshape(F, 3, 3), L.f32(To).reshape(F, 3)
        z = L.f32(z).reshape(N, S)
        sample_dist = float(torch.tensor((ctx.far - ctx.near) / ren.n_samples, dtype=torch.float32))
        sdf_h, sdf_o = L.f32(sdf_h).reshape(n), L.f32(sdf_o).reshape(n)
        grad_h, grad_o = L.f32(grad_h).reshape(n, 3), L.f32(grad_o).reshape(n, 3)
        tape = ctx.tape if (ctx.tape is not None and ctx.tape_serial == getattr(ren, '_tape_serial', -1)) else None
        if getattr(ren, '_backward_depths', None) is not None:
            tape = None
            # test hook: differentiate on GIVEN depths -- the per-sample values are evaluated there first
            z = L.f32(ren._backward_depths).reshape(N, S)

            def field_forward(field, o, d, frames):
                pts, dists = _empty(n, 3, dev=dev), _empty(n, dev=dev)
                L.check(lib.hn_sample_points(L.ptr(o), L.ptr(d), L.ptr(z), N, S, 1, sample_dist, L.ptr(pts), L.ptr(dists), st), 'hn_sample_points')
                sdf, grad, rgb = _empty(n, dev=dev), _empty(n, 3, dev=dev), _empty(n, 3, dev=dev)
                wsb = lib.hn_field_workspace_bytes(field.handle, n)
                ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
                L.check(lib.hn_field_eval(field.handle, L.ptr(pts), L.ptr(d), n, S, L.ptr(bt) if frames else None,
                                          L.ptr(tp) if frames else None, F if frames else 1, P * S if frames else n, L.ptr(sdf),
                                          L.ptr(grad), L.ptr(rgb), None, L.ptr(ws), wsb, st), 'hn_field_eval')
                alpha = _empty(n, dev=dev)
                L.check(lib.hn_alpha(L.ptr(sdf), L.ptr(grad), L.ptr(d), L.ptr(dists), n, S, float(field.inv_s), L.ptr(alpha), None, st), 'hn_alpha')
                return sdf, grad, rgb, alpha

            o_l, d_l = _empty(N, 3, dev=dev), _empty(N, 3, dev=dev)
            L.check(lib.hn_obj_local_fwd(L.ptr(ro), L.ptr(rd), L.ptr(Ro), L.ptr(To), F, P, L.ptr(o_l), L.ptr(d_l), st), 'hn_obj_local_fwd')
            sdf_h, grad_h, rgb_h, alpha_h = field_forward(hand, ro, rd, True)
            sdf_o, grad_o, rgb_o, alpha_o = field_forward(obj, o_l, d_l, False)
        opt = lambda g, shape: None if g is None else L.f32(g).reshape(shape)
        g_color = L.f32(g_color).reshape(N, 3) if g_color is not None else torch.zeros(N, 3, device=dev)
        ups = (opt(g_wsum, (N,)), opt(g_sdf_h, (n,)), opt(g_sdf_o, (n,)), opt(g_grad_h, (n, 3)), opt(g_grad_o, (n, 3)), opt(g_gerr, (2,)))
        # d loss / d the world rays only when a caller differentiates w.r.t. them (the fitting loops' rays come from fixed cameras)
        want_rays = ctx.want_rays
        g_ro = _empty(N, 3, dev=dev) if want_rays else None
        g_rd = _empty(N, 3, dev=dev) if want_rays else None
        g_bt, g_tp = _empty(F, 21, 4, 4, dev=dev), _empty(F, 21, 3, dev=dev)
        g_Ro, g_To = _empty(F, 3, 3, dev=dev), _empty(F, 3, dev=dev)
        need = lib.hn_render_dual_bwd_workspace_bytes(hand.handle, obj.handle, N, S)
        ws = ren._ws_bwd.get(need, dev)          # grow-only, re-used across steps
        L.check(lib.hn_render_dual_bwd(hand.handle, obj.handle, L.ptr(ro), L.ptr(rd), F, P, S, sample_dist, L.ptr(bt), L.ptr(tp), L.ptr(Ro),
                                       L.ptr(To), L.ptr(z), L.ptr(sdf_h), L.ptr(grad_h), L.ptr(rgb_h), L.ptr(alpha_h),
                                       L.ptr(sdf_o), L.ptr(grad_o), L.ptr(rgb_o), L.ptr(alpha_o), L.ptr(g_color),
                                       L.ptr(ups[0]), L.ptr(ups[1]), L.ptr(ups[2]), L.ptr(ups[3]), L.ptr(ups[4]), L.ptr(ups[5]),
                                       L.ptr(g_ro), L.ptr(g_rd), L.ptr(g_bt), L.ptr(g_tp), L.ptr(g_Ro), L.ptr(g_To), L.ptr(ws), need,
                                       L.ptr(tape), 0, st),
                'hn_render_dual_bwd')

        _join_pending_side()       # (the stable term's backward pass, queued on its own stream by the loss node: see _PENDING_SIDE)

        def like(g, ref):          # an input shared by all frames (e.g. T_pose [21,3]) receives the sum over frames
            return g.reshape(ref.shape) if g.numel() == ref.numel() else g.reshape(F, *ref.shape[-(ref.dim()):]).sum(0).reshape(ref.shape)

        sv = ctx.saved_tensors
        return (g_ro.reshape(rays_o.shape) if want_rays else None, g_rd.reshape(rays_d.shape) if want_rays else None, like(g_bt, sv[2]),
                like(g_tp, sv[3]), like(g_Ro, sv[4]), like(g_To, sv[5]), None, None, None, None)


class HandSdfFn(torch.autograd.Function):
    """`sdf_network_hand.sdf(pts, bt_inv, T_pose_21)` with gradients into the points and the bone transforms: the hand
    SDF on the object's vertices in get_stable_loss_cross (utils/renderer_batch.py:318-371, back-propagated by
    fitting_video.py:340-342).  pts [F,M,3], bt_inv [F,21,4,4], T_pose [F,21,3] -> sdf [F,M].  Forward = one
    hn_field_sdf launch; backward = hn_field_eval_bwd in its sdf-only form (g_grad = g_rgb = NULL)."""

    @staticmethod
    def forward(ctx, pts, bt_inv, T_pose, field, ws):
        L = _lib
        lib = L.load()
        F, M = pts.shape[0], pts.shape[1]
        p = L.f32(pts).reshape(F * M, 3)
        bt, tp = L.f32(bt_inv).reshape(F, 21, 4, 4), L.f32(T_pose).reshape(-1, 21, 3)
        if tp.shape[0] != F:
            tp = tp.expand(F, 21, 3).contiguous()
        n = F * M
        out = _empty(n, dev=p.device)
        need = lib.hn_field_workspace_bytes(field.handle, n)
        buf = ws.get(max(need, 16), p.device)
        L.check(lib.hn_field_sdf(field.handle, L.ptr(p), n, L.ptr(bt), L.ptr(tp), F, M, L.ptr(out), L.ptr(buf), need, L.stream_ptr()),
                'hn_field_sdf')
        ctx.field, ctx.ws, ctx.shapes = field, ws, (pts.shape, bt_inv.shape, T_pose.shape)
        ctx.save_for_backward(p, bt, tp)
        return out.reshape(F, M)

    @staticmethod
    def backward(ctx, g_sdf):
        L = _lib
        lib = L.load()
        p, bt, tp = ctx.saved_tensors
        F = bt.shape[0]
        n = p.shape[0]
        dev = p.device
        gs = L.f32(g_sdf).reshape(n)
        g_pts = _empty(n, 3, dev=dev)
        g_bt, g_tp = torch.zeros(F, 21, 4, 4, device=dev), torch.zeros(F, 21, 3, device=dev)
        need = lib.hn_field_bwd_workspace_bytes(ctx.field.handle, n)
        buf = ctx.ws.get(need, dev)
        L.check(lib.hn_field_eval_bwd(ctx.field.handle, L.ptr(p), None, n, 1, L.ptr(bt), L.ptr(tp), F, n // F, L.ptr(gs), None, None,
                                      L.ptr(g_pts), None, L.ptr(g_bt), L.ptr(g_tp), L.ptr(buf), need, L.stream_ptr()),
                'hn_field_eval_bwd')
        sp, sb, st = ctx.shapes
        g_tp = g_tp.reshape(st) if g_tp.numel() == int(torch.Size(st).numel()) else g_tp.sum(0).reshape(st)
        return g_pts.reshape(sp), g_bt.reshape(sb), g_tp, None, None


class FitLossFn(torch.autograd.Function):
    """The four render-dependent loss terms of a fitting step (fitting_single.py:251-283) as two launches
    (hn_fit_loss_sums / hn_fit_loss_grads): (color_fine [R,3], weight_sum [R,1], sdf_hand [n,1] | None, sdf_obj | None,
    true_rgb, true_mask) -> (colour loss, mask loss, contact, penetration); the last two are 0 without sdfs."""

    @staticmethod
    def forward(ctx, color, wsum, sdf_h, sdf_o, true_rgb, true_mask):
        L = _lib
        lib = L.load()
        dev = color.device
        c, w = L.f32(color).reshape(-1, 3), L.f32(wsum).reshape(-1)
        t, m = L.f32(true_rgb, dev).reshape(-1, 3), L.f32(true_mask, dev).reshape(-1)
        sh = None if sdf_h is None else L.f32(sdf_h).reshape(-1)
        so = None if sdf_o is None else L.f32(sdf_o).reshape(-1)
        R, n = c.shape[0], 0 if sh is None else sh.shape[0]
        sums = _empty(6, dev=dev)
        L.check(lib.hn_fit_loss_sums(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), R, L.ptr(sh), L.ptr(so), n, L.ptr(sums), L.stream_ptr()),
                'hn_fit_loss_sums')
        ctx.save_for_backward(c, w, t, m, sums, *([sh, so] if sh is not None else []))
        ctx.shapes = (color.shape, wsum.shape, None if sdf_h is None else sdf_h.shape)
        # views and two divisions: nothing here creates a tensor from host data (that would be a synchronising copy)
        return sums[0], sums[1], sums[2] / (sums[3] + 1e-9), sums[4] / (sums[5] + 1e-9)

    @staticmethod
    def backward(ctx, g_c, g_m, g_ct, g_p):
        L = _lib
        lib = L.load()
        sv = ctx.saved_tensors
        c, w, t, m, sums = sv[:5]
        sh, so = (sv[5], sv[6]) if len(sv) > 5 else (None, None)
        dev = c.device
        g4 = torch.zeros(4, device=dev)
        for k, gk in enumerate((g_c, g_m, g_ct, g_p)):
            if gk is not None:
                g4[k] = gk
        R, n = c.shape[0], 0 if sh is None else sh.shape[0]
        gc, gw = _empty(R, 3, dev=dev), _empty(R, dev=dev)
        gsh = _empty(n, dev=dev) if sh is not None else None
        gso = _empty(n, dev=dev) if sh is not None else None
        L.check(lib.hn_fit_loss_grads(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), R, L.ptr(sh), L.ptr(so), n, L.ptr(sums), L.ptr(g4), L.ptr(gc),
                                      L.ptr(gw), L.ptr(gsh), L.ptr(gso), L.stream_ptr()), 'hn_fit_loss_grads')
        s_c, s_w, s_s = ctx.shapes
        return (gc.reshape(s_c), gw.reshape(s_w), None if gsh is None else gsh.reshape(s_s), None if gso is None else gso.reshape(s_s),
                None, None)


_LOSS_SCRATCH = {}


def _loss_scratch(lib, n_rays, n_samples, dev, frames=1):
    """The partial-sum slots + counter of hn_fit_step_loss: zeroed once, every launch leaves it ready for the next.  One per
    (device, stream): steps of two streams must not share a counter.  frames > 1: that many frames' slots behind one another
    (hn_fit_step_loss_frames) -- a buffer of its own per layout, because frame f's counter sits where another layout keeps sums."""
    per_frame = lib.hn_fit_step_loss_scratch_bytes(n_rays, n_samples)
    need = max(per_frame * int(frames), lib.hn_window_loss_scratch_bytes(n_rays, n_samples))   # (one buffer serves both)
    key = (str(dev), torch.cuda.current_stream().cuda_stream) + ((int(frames), int(per_frame)) if frames > 1 else ())
    buf = _LOSS_SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        buf = _LOSS_SCRATCH[key] = torch.zeros(int(need), dtype=torch.uint8, device=dev)
    return buf, need


class FitStepLossFn(torch.autograd.Function):
    """The whole loss of one fitting_single step (fitting_single.py:251-288) as ONE launch forward (hn_fit_step_loss: the sums of
    the render terms, the vertex loss, the joint loss, the weighted total) and ONE backward (hn_fit_step_loss_bwd), instead of
    ~25 + ~35 element-wise torch launches in a step that is a chain of dependent launches:
      (color_fine [R,3], weight_sum [R,1], sdf_hand [n,1] | None, sdf_obj | None, joint_3d [1,21,3], obj_r [1,3,3], obj_t [1,3])
      -> (loss [], terms [8] = loss, colour, mask, contact, penetration, joint, verts, 0  -- not differentiable).
    weights = (w_render, w_contact, w_penetration, w_joint, w_verts)."""

    @staticmethod
    def forward(ctx, color, wsum, sdf_h, sdf_o, joint_3d, obj_r, obj_t, true_rgb, true_mask, joint_pred, Ro_pred, To_pred, verts, weights):
        L = _lib
        lib = L.load()
        dev = color.device
        st = L.stream_ptr()
        c, w = L.f32(color).reshape(-1, 3), L.f32(wsum).reshape(-1)
        t, m = L.f32(true_rgb, dev).reshape(-1, 3), L.f32(true_mask, dev).reshape(-1)
        sh = None if sdf_h is None else L.f32(sdf_h).reshape(-1)
        so = None if sdf_o is None else L.f32(sdf_o).reshape(-1)
        R, n = c.shape[0], 0 if sh is None else sh.shape[0]
        j3, jp = L.f32(joint_3d).reshape(-1, 3), L.f32(joint_pred, dev).reshape(-1, 3)
        nj = j3.shape[0]
        assert jp.shape[0] == nj and nj <= 64, 'one frame of joints'
        Ra, ta = L.f32(obj_r).reshape(1, 9), L.f32(obj_t).reshape(1, 3)
        Rb, tb = L.f32(Ro_pred, dev).reshape(1, 9), L.f32(To_pred, dev).reshape(1, 3)
        buf = _empty(6 + 1 + 9 + 3 + 8 + 3 * nj, dev=dev)          # one allocation: sums | (unused) | gR | gt | terms | g_joint
        sums, gR, gt, terms, gj = buf[0:6], buf[7:16], buf[16:19], buf[19:27], buf[27:27 + 3 * nj]
        scratch, need = _loss_scratch(lib, R, n, dev)
        w5 = (ctypes.c_float * 5)(*[float(x) for x in weights])
        L.check(lib.hn_fit_step_loss(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), R, L.ptr(sh), L.ptr(so), n, L.ptr(j3), L.ptr(jp), nj, L.ptr(Ra), L.ptr(ta),
                                     L.ptr(Rb), L.ptr(tb), L.ptr(verts), verts.shape[0], w5, L.ptr(scratch), need, L.ptr(sums), L.ptr(terms), L.ptr(gj),
                                     L.ptr(gR), L.ptr(gt), st), 'hn_fit_step_loss')
        ctx.save_for_backward(c, w, t, m, buf, *([sh, so] if sh is not None else []))
        ctx.w5, ctx.nj = w5, nj
        ctx.shapes = (color.shape, wsum.shape, None if sdf_h is None else sdf_h.shape, joint_3d.shape, obj_r.shape, obj_t.shape)
        ctx.mark_non_differentiable(terms)
        return terms[0], terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        L = _lib
        lib = L.load()
        sv = ctx.saved_tensors
        c, w, t, m, buf = sv[:5]
        sh, so = (sv[5], sv[6]) if len(sv) > 5 else (None, None)
        nj = ctx.nj
        sums, gR, gt, gj = buf[0:6], buf[7:16], buf[16:19], buf[27:27 + 3 * nj]
        dev = c.device
        R, n = c.shape[0], 0 if sh is None else sh.shape[0]
        out = _empty(3 * nj + 9 + 3, dev=dev)
        gj_o, gR_o, gt_o = out[0:3 * nj], out[3 * nj:9 + 3 * nj], out[9 + 3 * nj:12 + 3 * nj]
        gl = L.f32(g_loss).reshape(1)
        gc, gw = _empty(R, 3, dev=dev), _empty(R, dev=dev)
        gsh = _empty(n, dev=dev) if sh is not None else None
        gso = _empty(n, dev=dev) if sh is not None else None
        L.check(lib.hn_fit_step_loss_bwd(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), R, L.ptr(sh), L.ptr(so), n, L.ptr(sums), L.ptr(gl), ctx.w5, L.ptr(gj),
                                         L.ptr(gR), L.ptr(gt), nj, L.ptr(gc), L.ptr(gw), L.ptr(gsh), L.ptr(gso), L.ptr(gj_o), L.ptr(gR_o), L.ptr(gt_o),
                                         L.stream_ptr()), 'hn_fit_step_loss_bwd')
        s_c, s_w, s_s, s_j, s_r, s_t = ctx.shapes
        return (gc.reshape(s_c), gw.reshape(s_w), None if gsh is None else gsh.reshape(s_s), None if gso is None else gso.reshape(s_s),
                gj_o.reshape(s_j), gR_o.reshape(s_r), gt_o.reshape(s_t), None, None, None, None, None, None, None)


class Mat3InverseFn(torch.autograd.Function):
    """torch.inverse on [F,3,3] (fitting_video.py:284 `torch.inverse(obj_r)`) as one launch each way (hn_mat3_inverse / _bwd); as
    torch operators the LU factorisation, the solve and their backward passes were ~15 launches of a launch-bound step."""

    @staticmethod
    def forward(ctx, R):
        L = _lib
        lib = L.load()
        r = L.f32(R).reshape(-1, 9)
        out = torch.empty_like(r)
        L.check(lib.hn_mat3_inverse(L.ptr(r), r.shape[0], L.ptr(out), L.stream_ptr()), 'hn_mat3_inverse')
        ctx.save_for_backward(out)
        ctx.shape = R.shape
        return out.reshape(R.shape)

    @staticmethod
    def backward(ctx, g):
        L = _lib
        lib = L.load()
        (y,) = ctx.saved_tensors
        _join_pending_side(y.device)        # (g may carry the stable term's share, produced on its own stream: see _PENDING_SIDE)
        gy = L.f32(g).reshape(-1, 9)
        gr = torch.empty_like(y)
        L.check(lib.hn_mat3_inverse_bwd(L.ptr(y), L.ptr(gy), y.shape[0], L.ptr(gr), L.stream_ptr()), 'hn_mat3_inverse_bwd')
        return gr.reshape(ctx.shape)


_DIAG, _DIAG_CHECK = [], [False]


class StableTerm:
    """`get_stable_loss_cross` (utils/renderer_batch.py:318-371) in explicit halves, no autograd: `StableTerm(...)` runs the forward
    -- hn_stable_pts (every 10th vertex to the world), the hand field's TAPED evaluation on those points, hn_stable_value (inside sets,
    nearest outside vertices, weights, the value and d value / d sdf): three launches -- and `.backward(g)` the rest: the hand field's
    adjoint from the tape with g x d value / d sdf as the upstream gradient of the sdf, then hn_stable_pts_bwd
    -> (g_bt_inv, g_T_pose, g_obj_r, g_obj_t).  (As torch operators around a layer-by-layer sdf adjoint: ~50 + ~90 launches.)
    `StableLossFn` wraps it as an autograd node; the fitting_video loss node (`FitWindowLossFn`) drives it directly, so that its
    backward launches are queued FIRST in a step's backward pass."""

    def __init__(self, pts, bt_inv, T_pose, obj_r, obj_t, field, state, strict):
        L = _lib
        lib = L.load()
        dev = torch.device('cuda')
        st = L.stream_ptr()
        p = L.f32(pts, dev)
        Fr, Vfull = p.shape[0], p.shape[1]
        stride = 10
        V = (Vfull + stride - 1) // stride
        n = Fr * V
        bt, tp = L.f32(bt_inv, dev).reshape(Fr, 21, 4, 4), L.f32(T_pose, dev).reshape(-1, 21, 3)
        if tp.shape[0] != Fr:
            tp = tp.expand(Fr, 21, 3).contiguous()
        R, t = L.f32(obj_r, dev).reshape(Fr, 9), L.f32(obj_t, dev).reshape(Fr, 3)
        pw, p0 = _empty(n, 3, dev=dev), _empty(V, 3, dev=dev)
        L.check(lib.hn_stable_pts(L.ptr(p), Fr, Vfull, stride, L.ptr(R), L.ptr(t), L.ptr(pw), L.ptr(p0), st), 'hn_stable_pts')
        if _DIAG_CHECK[0]:       # (tools/seq_repro_diag.py: what this stream sees of its inputs right behind the launch, kept for the caller to compare)
            ref = torch.einsum('frc,fvc->fvr', R.view(Fr, 3, 3), p[:, ::stride]) + t[:, None]
            _DIAG.append({'pw_vs_ref_on_side': (pw.view(Fr, V, 3) - ref).abs().max().reshape(1), 'R_side': R.clone(), 't_side': t.clone(),
                          'obj_r': obj_r, 'obj_t': obj_t})
        sdf, grad, rgb = _empty(n, dev=dev), _empty(n, 3, dev=dev), _empty(n, 3, dev=dev)
        tape_bytes = lib.hn_field_tape_bytes(field.handle, n)
        need = lib.hn_field_workspace_bytes(field.handle, n)
        tape = state['tape'].get(max(tape_bytes, 16), dev)
        ws = state['ws'].get(max(need, 16), dev)
        dirs = state.setdefault('dirs', {})
        if n not in dirs:
            dirs[n] = torch.zeros(n, 3, device=dev)          # (the hand's colour network ignores the view direction; also the zero upstream gradients)
        L.check(lib.hn_field_eval_taped(field.handle, L.ptr(pw), L.ptr(dirs[n]), n, 1, L.ptr(bt), L.ptr(tp), Fr, V, L.ptr(sdf), L.ptr(grad), L.ptr(rgb),
                                        L.ptr(ws), need, L.ptr(tape), tape_bytes, st), 'hn_field_eval_taped')
        val_d = _empty(1 + n, dev=dev)
        value, dsdf = val_d[0:1], val_d[1:]
        sneed = lib.hn_stable_value_scratch_bytes(Fr, V)
        scr = state.get('value_scratch')
        if scr is None or scr.numel() < sneed:
            scr = state['value_scratch'] = torch.zeros(int(sneed), dtype=torch.uint8, device=dev)     # zeroed once: every launch leaves its counter zero
        L.check(lib.hn_stable_value(L.ptr(sdf), L.ptr(p0), Fr, V, 1 if strict else 0, L.ptr(value), L.ptr(dsdf), L.ptr(scr), sneed, st), 'hn_stable_value')
        state['serial'] = state.get('serial', 0) + 1
        self.serial, self.state, self.field = state['serial'], state, field
        self.sizes = (Fr, Vfull, stride, V)
        self.saved = (p, pw, bt, tp, dsdf, grad, rgb)
        self.value = value.reshape(())
        self.stream = torch.cuda.current_stream()

    def backward(self, g):
        """g: upstream gradient of the value (device scalar) -> (g_bt_inv [F,21,4,4], g_T_pose [F,21,3], g_obj_r [F,9], g_obj_t [F,3])."""
        L = _lib
        lib = L.load()
        p, pw, bt, tp, dsdf, grad, rgb = self.saved
        if self.serial != self.state.get('serial'):
            raise RuntimeError('the stable term\'s tape was overwritten by a later evaluation before its backward pass ran')
        Fr, Vfull, stride, V = self.sizes
        n = Fr * V
        dev = p.device
        st = L.stream_ptr()
        gs = dsdf * g                                           # upstream gradient of the sdf
        zeros = self.state['dirs'][n]
        g_pts = _empty(n, 3, dev=dev)
        g_pose = torch.zeros(Fr * (21 * 16 + 21 * 3), device=dev)  # (one fill: the adjoint accumulates into both blocks)
        g_bt_c, g_tp_c = g_pose[:Fr * 336].view(Fr, 21, 4, 4), g_pose[Fr * 336:].view(Fr, 21, 3)
        need = lib.hn_field_bwd_workspace_bytes(self.field.handle, n)
        ws = self.state['ws_bwd'].get(max(need, 16), dev)
        tape = self.state['tape'].buf
        L.check(lib.hn_field_eval_bwd_taped(self.field.handle, L.ptr(pw), L.ptr(zeros), n, 1, L.ptr(bt), L.ptr(tp), Fr, V, L.ptr(gs), L.ptr(zeros),
                                            L.ptr(zeros), L.ptr(grad), L.ptr(rgb), L.ptr(tape), L.ptr(g_pts), None, L.ptr(g_bt_c), L.ptr(g_tp_c), L.ptr(ws),
                                            need, st), 'hn_field_eval_bwd_taped')
        gR, gt = _empty(Fr, 9, dev=dev), _empty(Fr, 3, dev=dev)
        L.check(lib.hn_stable_pts_bwd(L.ptr(p), Fr, Vfull, stride, L.ptr(g_pts), L.ptr(gR), L.ptr(gt), st), 'hn_stable_pts_bwd')
        return g_bt_c, g_tp_c, gR, gt


class StableLossFn(torch.autograd.Function):
    """`get_stable_loss_cross` as one autograd node over StableTerm: (obj verts [F,Vfull,3], bt_inv [F,21,4,4], T_pose [F,21,3], obj_r
    [F,3,3], obj_t [F,3]) -> the stable term."""

    @staticmethod
    def forward(ctx, pts, bt_inv, T_pose, obj_r, obj_t, field, state, strict):
        ctx.term = StableTerm(pts, bt_inv, T_pose, obj_r, obj_t, field, state, strict)
        ctx.shapes = (bt_inv.shape, T_pose.shape, obj_r.shape, obj_t.shape)
        return ctx.term.value

    @staticmethod
    def backward(ctx, g):
        g_bt, g_tp, gR, gt = ctx.term.backward(g)
        s_bt, s_tp, s_r, s_t = ctx.shapes
        g_tp_out = g_tp.reshape(s_tp) if g_tp.numel() == int(torch.Size(s_tp).numel()) else g_tp.sum(0).reshape(s_tp)
        return None, g_bt.reshape(s_bt), g_tp_out, gR.reshape(s_r), gt.reshape(s_t), None, None, None


# streams whose tail a step's main stream has still to wait for before gradients produced there are consumed: the stable term's
# backward launches (FitWindowLossFn.backward, on the step's side stream); DualRenderFn.backward inserts the wait behind its own
# launches -- the render's adjoint kernels are then already queued, and the wait costs nothing (they run longer)
# Keyed per device; every node that may be the first consumer of those gradients drains its device's list (DualRenderFn, and --
# for a loss used without the render's backward, detached render inputs, an exception in between -- the pose-side nodes
# HaloChainFn / Mat3InverseFn, which receive them whatever else ran): a join is an event wait, and joining twice is free.
_PENDING_SIDE = {}
_PENDING_LOCK = threading.Lock()


def _push_pending_side(stream):
    with _PENDING_LOCK:
        _PENDING_SIDE.setdefault(stream.device.index, []).append(stream)


def _join_pending_side(device=None):
    cur = torch.cuda.current_stream(device)
    with _PENDING_LOCK:
        todo = _PENDING_SIDE.pop(cur.device.index, [])
    for st in todo:
        cur.wait_stream(st)


class FitWindowLossFn(torch.autograd.Function):
    """The whole loss of one fitting_video window step (fitting_video.py:285-334) as ONE launch forward (hn_window_loss) and ONE
    backward (hn_window_loss_bwd): (color_fine [F,P,3], weight_sum [F,P,1], sdf_hand [n,1], sdf_obj [n,1], joint_3d [F,21,3], obj_r
    [F,3,3], obj_t [F,3], stable [] | None) -> (loss [], terms [10] = loss, colour, mask, contact, penetration, joint, verts,
    50 x smooth, 100 x stable, 0 -- not differentiable).  anchor: 1 / 2 = the window starts / ends the sequence (not on the very
    first step).  As torch operators (render terms, pose regularisers, smoothness): ~60 launches forward, ~80 backward."""

    WEIGHTS = (0.5, 30.0, 20.0, 30.0, 20.0, 50.0, 100.0)

    @staticmethod
    def forward(ctx, color, wsum, sdf_h, sdf_o, joint_3d, obj_r, obj_t, stable, true_rgb, true_mask, joint_pred, Ro_pred, To_pred, verts, anchor,
                bt_inv=None, stable_term=None):
        """stable: the stable term as a tensor (an autograd input), or None with `stable_term`: a StableTerm evaluated earlier on
        `stable_term.stream` (no autograd) whose backward this node's backward drives -- its gradients w.r.t. bt_inv, obj_r, obj_t
        are returned with this node's (bt_inv is an input for that purpose only)."""
        L = _lib
        lib = L.load()
        dev = color.device
        ctx.stable_term = stable_term
        if stable_term is not None:
            assert stable is None
            stable = stable_term.value
        c, w = L.f32(color).reshape(-1, 3), L.f32(wsum).reshape(-1)
        t, m = L.f32(true_rgb, dev).reshape(-1, 3), L.f32(true_mask, dev).reshape(-1)
        sh, so = L.f32(sdf_h).reshape(-1), L.f32(sdf_o).reshape(-1)
        R, n = c.shape[0], sh.shape[0]
        j3, jp = L.f32(joint_3d).reshape(-1, 21, 3), L.f32(joint_pred, dev).reshape(-1, 21, 3)
        Fr = j3.shape[0]
        Ra, ta = L.f32(obj_r).reshape(Fr, 9), L.f32(obj_t).reshape(Fr, 3)
        Rb, tb = L.f32(Ro_pred, dev).reshape(Fr, 9), L.f32(To_pred, dev).reshape(Fr, 3)
        st_ = None if stable is None else L.f32(stable).reshape(1)
        buf = _empty(6 + 10 + 75 * Fr, dev=dev)          # sums | terms | g_joint [F,63] | gR [F,9] | gt [F,3]
        sums, terms = buf[0:6], buf[6:16]
        gj, gR, gt = buf[16:16 + 63 * Fr], buf[16 + 63 * Fr:16 + 72 * Fr], buf[16 + 72 * Fr:16 + 75 * Fr]
        scratch, need = _loss_scratch(lib, R, n, dev)
        w7 = (ctypes.c_float * 7)(*FitWindowLossFn.WEIGHTS)
        L.check(lib.hn_window_loss(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), R, L.ptr(sh), L.ptr(so), n, L.ptr(j3), L.ptr(jp), Fr, L.ptr(Ra), L.ptr(ta),
                                   L.ptr(Rb), L.ptr(tb), L.ptr(verts), verts.shape[0], L.ptr(st_), int(anchor), w7, L.ptr(scratch), need, L.ptr(sums),
                                   L.ptr(terms), L.ptr(gj), L.ptr(gR), L.ptr(gt), L.stream_ptr()), 'hn_window_loss')
        ctx.save_for_backward(c, w, t, m, buf, sh, so)
        ctx.w7, ctx.Fr, ctx.has_stable = w7, Fr, stable is not None
        ctx.shapes = (color.shape, wsum.shape, sdf_h.shape, sdf_o.shape, joint_3d.shape, obj_r.shape, obj_t.shape, None if bt_inv is None else bt_inv.shape)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)       # (no zero tensor for the gradient of `terms`)
        return terms[0], terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        if g_loss is None:
            return (None,) * 17
        L = _lib
        lib = L.load()
        c, w, t, m, buf, sh, so = ctx.saved_tensors
        Fr = ctx.Fr
        sums = buf[0:6]
        gj, gR, gt = buf[16:16 + 63 * Fr], buf[16 + 63 * Fr:16 + 72 * Fr], buf[16 + 72 * Fr:16 + 75 * Fr]
        dev = c.device
        R, n = c.shape[0], sh.shape[0]
        out = _empty(75 * Fr + 1, dev=dev)
        gj_o, gR_o, gt_o, gst = out[0:63 * Fr], out[63 * Fr:72 * Fr], out[72 * Fr:75 * Fr], out[75 * Fr:75 * Fr + 1]
        gl = L.f32(g_loss).reshape(1)
        gc, gw, gsh, gso = _empty(R, 3, dev=dev), _empty(R, dev=dev), _empty(n, dev=dev), _empty(n, dev=dev)
        L.check(lib.hn_window_loss_bwd(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), R, L.ptr(sh), L.ptr(so), n, L.ptr(sums), L.ptr(gl), ctx.w7, L.ptr(gj),
                                       L.ptr(gR), L.ptr(gt), Fr, L.ptr(gc), L.ptr(gw), L.ptr(gsh), L.ptr(gso), L.ptr(gj_o), L.ptr(gR_o), L.ptr(gt_o),
                                       L.ptr(gst), L.stream_ptr()), 'hn_window_loss_bwd')
        s_c, s_w, s_h, s_o, s_j, s_r, s_t, s_b = ctx.shapes
        g_bt = None
        term = ctx.stable_term
        if term is not None:
            # the stable term's backward pass, on the stream its forward ran on, queued NOW -- ahead of the render's adjoint kernels,
            # which take every CU for a millisecond (queued behind them its three small launches waited ~0.3 ms for a free CU and its
            # 7-tile adjoint ended ~0.35 ms after the render's).  The main stream waits for it behind the render's launches
            # (_PENDING_SIDE), before anything adds these gradients to the render's.
            cur = torch.cuda.current_stream()
            side = term.stream
            if side != cur:
                side.wait_stream(cur)
            with torch.cuda.stream(side):
                gb, _gtp, gRs, gts = term.backward(gst)
                gRt = gR_o.reshape(Fr, 9) + gRs
                gtt = gt_o.reshape(Fr, 3) + gts
            if side != cur:
                for x in (gb, gRt, gtt):
                    x.record_stream(cur)
                _push_pending_side(side)
            gR_o, gt_o, g_bt = gRt, gtt, (gb.reshape(s_b) if s_b is not None else None)
        return (gc.reshape(s_c), gw.reshape(s_w), gsh.reshape(s_h), gso.reshape(s_o), gj_o.reshape(s_j), gR_o.reshape(s_r), gt_o.reshape(s_t),
                gst.reshape(()) if (ctx.has_stable and term is None) else None, None, None, None, None, None, None, None, g_bt, None)

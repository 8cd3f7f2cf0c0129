"""Offline-stage training of one field over the C ABI: `exp_runner.train`'s
`render -> loss -> loss.backward() -> optimizer.step()` (exp_runner.py:126-242) with the renderer's HIP path on both
sides of the loss.

The reference differentiates `NeuSRenderer.render` w.r.t. every parameter of `sdf_network`, `color_network` and
`deviation_network` (Adam over them, exp_runner.py:97-104).  Sampling is under `no_grad` (utils/renderer.py:215), so the
backward pass runs through `render_core` at the final depths only: here one call, `hn_render_single_bwd`, which returns

  * d loss / d (folded weights W_l = g_l v_l / |v_l|, biases) of both networks as one flat vector in the layout of
    `hn_field_param_offset` (first-order path and the path through `.gradient()`),
  * d loss / d inv_s (inv_s = exp(10 variance)),
  * d loss / d rays (field frame), d loss / d bt_inv, T_pose (hand).

The weight-norm chain rule (d/d weight_g, d/d weight_v from d/d W; old-style `nn.utils.weight_norm`, dim 0) is one more
call (`hn_weight_norm_bwd`, all 14 layers); what is left for the host is the optimiser (torch's Adam, as the reference).  The VGG term of exp_runner.py:213-224 stays a torch module on `color_fine` ("gated on VGG loss
staying in torch"); it composes with this Function through autograd like any other loss on the render outputs.
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

from . import lib as _lib


def trainable_parameters(renderer):
    """The parameters `exp_runner` hands to Adam (exp_runner.py:97-103), in the fixed order SingleRenderFn uses:
    per SDF layer (weight_g, weight_v, bias), per colour layer the same, then `variance`.  (`se3_refine`, the other
    parameter of sdf_network, enters the render through `Ro / To` or `bt_inv` and receives its gradient from there.)"""
    ps = []
    for net in (renderer.sdf_network, renderer.color_network):
        for lin in net.layers():
            ps += [lin.weight_g, lin.weight_v, lin.bias]
    ps.append(renderer.deviation_network.variance)
    return ps


def _layer_slots(lib, field):
    """[(net, layer, w_off, b_off, out, in, ld)] of a packed field (hn_field_param_offset)."""
    slots = []
    for net, n_layers in ((0, 9), (1, 5)):
        for l in range(n_layers):
            w, b = ctypes.c_size_t(), ctypes.c_size_t()
            o, i, ld = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _lib.check(lib.hn_field_param_offset(field.handle, net, l, ctypes.byref(w), ctypes.byref(b), ctypes.byref(o),
                                                 ctypes.byref(i), ctypes.byref(ld)), 'hn_field_param_offset')
            slots.append((net, l, w.value, b.value, o.value, i.value, ld.value))
    return slots


def folded_gradients(lib, field, g_params):
    """Views of the flat gradient vector: [(dW [out,in], db [out])] in the order of `_layer_slots`."""
    out = []
    for _, _, w, b, o, i, ld in _layer_slots(lib, field):
        out.append((g_params[w:w + o * ld].view(o, ld)[:, :i], g_params[b:b + o]))
    return out


def weight_norm_backward(g, v, dW):
    """d/d weight_g [out,1], d/d weight_v [out,in] from d/d W for W = g v / |v| (row norms; torch's `_weight_norm`
    backward, utils/fields.py:113-121 `nn.utils.weight_norm(lin)`).  The torch statement of what hn_weight_norm_bwd
    computes on the device for all layers at once (kept for callers that hold folded gradients, and for the tests)."""
    nrm = v.norm(dim=1, keepdim=True)
    vh = v / nrm
    proj = (dW * vh).sum(dim=1, keepdim=True)
    return proj, (g / nrm) * (dW - vh * proj)


KEEP_TAPE = os.environ.get('HONERF_TRAIN_KEEP_TAPE', '1') != '0'   # the training render keeps its evaluation's tape for the backward pass
# ... when the block is no larger than this (41 KB per sample for the object nets, 42 KB per evaluated sample for the hand nets: 2.4 GB for
# the confs' 441 x 128 batch).  A differentiable render of a whole IMAGE (262 144 rays: 670 GB) goes through the plain pair, whose backward
# pass -- should one ever be asked for at that size -- evaluates the field again.
KEEP_TAPE_MAX_BYTES = int(float(os.environ.get('HONERF_TRAIN_KEEP_TAPE_MAX_GB', '24')) * (1 << 30))


class SingleRenderFn(torch.autograd.Function):
    """(rays_o [B,3], rays_d [B,3] in the FIELD's frame, bt_inv [21,4,4] | None, T_pose [21,3] | None, *parameters) ->
    (color_fine [B,3], weight_sum [B,1], gradient_error [], cdf_fine [B,S], weight_max [B,1]); the last two carry no
    gradient (they are logged only, exp_runner.py:237-238)."""

    @staticmethod
    def forward(ctx, renderer, near, far, t_rand, z_given, rays_o, rays_d, bt_inv, T_pose, *params):
        L = _lib
        lib = L.load()
        f = renderer.field()
        ro, rd = L.f32(rays_o).reshape(-1, 3), L.f32(rays_d).reshape(-1, 3)
        dev = ro.device
        B = ro.shape[0]
        S = renderer.n_samples + renderer.n_importance
        hand = f.kind == 'hand'
        tape = None
        bt = L.f32(bt_inv, dev).reshape(1, 21, 4, 4) if hand else None
        tp = L.f32(T_pose, dev).reshape(1, 21, 3) if hand else None
        sample_dist = float(torch.tensor((float(far) - float(near)) / renderer.n_samples, dtype=torch.float32))
        if z_given is not None:
            # render_core at the caller's depths (utils/renderer.py:107-177): the staged form of the same evaluation
            z = L.f32(z_given, dev).reshape(B, -1)
            S = z.shape[1]
            core = renderer.render_core(ro, rd, bt, tp, None, z, sample_dist)
            color, cdf, gerr = core['color'], core['cdf'], core['gradient_error'].reshape(1)
            wsum, wmax = core['weights'].sum(dim=-1, keepdim=True), core['weights'].max(dim=-1, keepdim=True)[0]
        else:
            from .renderer import _t_rand
            tr = _t_rand(t_rand, (B, 1), dev)
            color, cdf = torch.empty(B, 3, device=dev), torch.empty(B, S, device=dev)
            wsum, wmax = torch.empty(B, 1, device=dev), torch.empty(B, 1, device=dev)
            gerr, z = torch.empty(1, device=dev), torch.empty(B, S, device=dev)
            need = lib.hn_render_single_workspace_bytes(f.handle, B, renderer.n_samples, renderer.n_importance)
            ws = renderer._ws.get(need, dev)
            # The final evaluation keeps its tape for the backward pass where that pass takes one (an f16x3 field packed with its tape
            # programs: hn_render_single_tape_bytes > 0) -- the backward pass then does not evaluate the field a second time.
            # KEEP_TAPE = False: the plain pair (the same numbers to the bit; tests, A/B).
            tape_bytes = lib.hn_render_single_tape_bytes(f.handle, B, S) if (KEEP_TAPE and B > 0 and any(ctx.needs_input_grad)) else 0
            if tape_bytes > KEEP_TAPE_MAX_BYTES:
                tape_bytes = 0
            if tape_bytes:
                tape = torch.empty(tape_bytes, dtype=torch.uint8, device=dev)
                L.check(lib.hn_render_single_taped(f.handle, L.ptr(ro), L.ptr(rd), L.ptr(tr), B, float(near), float(far), renderer.n_samples,
                                                   renderer.n_importance, renderer.up_sample_steps, L.ptr(bt), L.ptr(tp), L.ptr(color),
                                                   L.ptr(cdf), L.ptr(wsum), L.ptr(wmax), L.ptr(gerr), L.ptr(z), L.ptr(tape), tape_bytes, L.ptr(ws),
                                                   ws.numel(), L.stream_ptr()), 'hn_render_single_taped')
            else:
                L.check(lib.hn_render_single(f.handle, L.ptr(ro), L.ptr(rd), L.ptr(tr), B, float(near), float(far), renderer.n_samples,
                                             renderer.n_importance, renderer.up_sample_steps, L.ptr(bt), L.ptr(tp), L.ptr(color),
                                             L.ptr(cdf), L.ptr(wsum), L.ptr(wmax), L.ptr(gerr), L.ptr(z), L.ptr(ws), ws.numel(),
                                             L.stream_ptr()), 'hn_render_single')
        renderer.last_z_vals = z
        ctx.renderer, ctx.field, ctx.S = renderer, f, S
        ctx.sample_dist = sample_dist
        ctx.n_params = len(params)
        ctx.tape = tape                # (a plain attribute: the block is this call's own, nothing else reads or versions it)
        ctx.tape_compact = bool(getattr(renderer, 'compact_far_field', False))   # the block's rows are the compacted list's when this is on
        ctx.save_for_backward(ro, rd, z, *([bt, tp] if hand else []))
        ctx.mark_non_differentiable(cdf, wmax)
        ctx.set_materialize_grads(False)       # (no zero tensors for outputs the loss does not use: [B,S] and [B,1] fills per iteration)
        return color, wsum, gerr.reshape(()), cdf, wmax

    @staticmethod
    def backward(ctx, g_color, g_wsum, g_gerr, _g_cdf, _g_wmax):
        L = _lib
        lib = L.load()
        ren, f, S = ctx.renderer, ctx.field, ctx.S
        sv = ctx.saved_tensors
        ro, rd, z = sv[:3]
        hand = f.kind == 'hand'
        bt, tp = (sv[3], sv[4]) if hand else (None, None)
        dev = ro.device
        B = ro.shape[0]
        gc = L.f32(g_color).reshape(B, 3) if g_color is not None else torch.zeros(B, 3, device=dev)
        gw = None if g_wsum is None else L.f32(g_wsum).reshape(B)
        ge = None if g_gerr is None else L.f32(g_gerr).reshape(1)
        if g_color is None and g_wsum is None and g_gerr is None:
            return (None,) * (9 + ctx.n_params)
        n_floats = lib.hn_field_param_floats(f.handle)
        zeros = torch.zeros(n_floats + 1, device=dev)        # (one fill for both)
        g_params, g_inv_s = zeros[:n_floats], zeros[n_floats:]
        g_ro, g_rd = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
        # pose gradients only when the caller differentiates the pose (exp_runner trains the networks on fixed poses: the bone maps'
        # kernels then skip 15 wave sums and atomics per sample block and bone)
        g_bt = torch.zeros(21, 4, 4, device=dev) if hand and ctx.needs_input_grad[7] else None
        g_tp = torch.zeros(21, 3, device=dev) if hand and ctx.needs_input_grad[8] else None
        need = lib.hn_render_single_bwd_workspace_bytes(f.handle, B, S)
        if not hasattr(ren, '_ws_train'):
            from .renderer import _Workspace
            ren._ws_train = _Workspace()
        ws = ren._ws_train.get(need, dev)
        tape = getattr(ctx, 'tape', None)
        if tape is not None and ctx.tape_compact != bool(getattr(ren, 'compact_far_field', False)):
            tape = None                # the far-field setting changed between the passes: the block's layout is the other one's -- re-evaluate
        if tape is not None:
            L.check(lib.hn_render_single_bwd_taped(f.handle, L.ptr(ro), L.ptr(rd), B, S, ctx.sample_dist, L.ptr(bt), L.ptr(tp), L.ptr(z), L.ptr(gc),
                                                   L.ptr(gw), L.ptr(ge), L.ptr(g_params), L.ptr(g_inv_s), L.ptr(g_ro), L.ptr(g_rd), L.ptr(g_bt),
                                                   L.ptr(g_tp), L.ptr(tape), tape.numel(), L.ptr(ws), need, L.stream_ptr()), 'hn_render_single_bwd_taped')
            ctx.tape = None            # (a second backward through the same graph -- retain_graph -- would need it: not offered, as the
            del tape                   #  reference's training loop does not; the block goes back to the allocator here)
        else:
            L.check(lib.hn_render_single_bwd(f.handle, L.ptr(ro), L.ptr(rd), B, S, ctx.sample_dist, L.ptr(bt), L.ptr(tp), L.ptr(z), L.ptr(gc),
                                             L.ptr(gw), L.ptr(ge), L.ptr(g_params), L.ptr(g_inv_s), L.ptr(g_ro), L.ptr(g_rd), L.ptr(g_bt),
                                             L.ptr(g_tp), L.ptr(ws), need, L.stream_ptr()), 'hn_render_single_bwd')
        # weight-norm chain rule of all 14 layers: one C-ABI call (hn_weight_norm_bwd), outputs in the parameter order of
        # trainable_parameters()
        from .nets import _mlp_desc
        keep = []
        d_sdf = _mlp_desc(ren.sdf_network.state_dict(), keep)
        d_col = _mlp_desc(ren.color_network.state_dict(), keep)
        grads, descs = [], []
        for net in (ren.sdf_network, ren.color_network):
            d = L.MlpDesc()
            layers = net.layers()
            d.n_layers = len(layers)
            for l, lin in enumerate(layers):
                dg, dv, db = torch.empty_like(lin.weight_g), torch.empty_like(lin.weight_v), torch.empty_like(lin.bias)
                d.weight_g[l], d.weight_v[l], d.bias[l] = dg.data_ptr(), dv.data_ptr(), db.data_ptr()
                d.out_dim[l], d.in_dim[l] = lin.weight_v.shape
                grads += [dg, dv, db]
            descs.append(d)
        L.check(lib.hn_weight_norm_bwd(f.handle, ctypes.byref(d_sdf), ctypes.byref(d_col), L.ptr(g_params), ctypes.byref(descs[0]),
                                       ctypes.byref(descs[1]), L.stream_ptr()), 'hn_weight_norm_bwd')
        del keep
        # inv_s = clip(exp(10 variance), 1e-6, 1e6) (utils/fields.py:248-249, utils/renderer.py:144)
        if f.inv_s_t is not None:       # the trained value never visited the host: the chain rule on the device too
            g_var = torch.empty(1, device=dev, dtype=torch.float32)
            L.check(lib.hn_variance_chain(L.ptr(g_inv_s), L.ptr(f.inv_s_t), L.ptr(g_var), L.stream_ptr()), 'hn_variance_chain')
            grads.append(g_var.reshape(()))
        else:
            inv_s = float(f.inv_s)
            grads.append((g_inv_s * (10.0 * inv_s if 1e-6 < inv_s < 1e6 else 0.0)).reshape(()))
        assert len(grads) == ctx.n_params, 'pass trainable_parameters(renderer) as the parameter list'
        return (None, None, None, None, None, g_ro, g_rd, g_bt, g_tp, *grads)


class ObjLocalFn(torch.autograd.Function):
    """convert_obj_to_local (utils/renderer.py:180-188): o' = Ro (o - To), d' = Ro d, with its adjoint
    (hn_obj_local_fwd / hn_obj_local_bwd) -- the path of the object's `se3_refine` leaves (exp_runner.py:155-161)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, Ro, To):
        L = _lib
        lib = L.load()
        ro, rd = L.f32(rays_o).reshape(-1, 3), L.f32(rays_d).reshape(-1, 3)
        R, T = L.f32(Ro, ro.device).reshape(1, 3, 3), L.f32(To, ro.device).reshape(1, 3)
        o2, d2 = torch.empty_like(ro), torch.empty_like(rd)
        L.check(lib.hn_obj_local_fwd(L.ptr(ro), L.ptr(rd), L.ptr(R), L.ptr(T), 1, ro.shape[0], L.ptr(o2), L.ptr(d2), L.stream_ptr()),
                'hn_obj_local_fwd')
        ctx.save_for_backward(ro, rd, R, T)
        ctx.shapes = (rays_o.shape, rays_d.shape, Ro.shape, To.shape)
        return o2, d2

    @staticmethod
    def backward(ctx, g_o, g_d):
        L = _lib
        lib = L.load()
        ro, rd, R, T = ctx.saved_tensors
        dev = ro.device
        go, gd = L.f32(g_o).reshape(-1, 3), L.f32(g_d).reshape(-1, 3)
        g_ro, g_rd = torch.empty_like(ro), torch.empty_like(rd)
        g_R, g_T = torch.empty(1, 3, 3, device=dev), torch.empty(1, 3, device=dev)
        L.check(lib.hn_obj_local_bwd(L.ptr(ro), L.ptr(rd), L.ptr(R), L.ptr(T), L.ptr(go), L.ptr(gd), 1, ro.shape[0], L.ptr(g_ro),
                                     L.ptr(g_rd), L.ptr(g_R), L.ptr(g_T), L.stream_ptr()), 'hn_obj_local_bwd')
        s = ctx.shapes
        return g_ro.reshape(s[0]), g_rd.reshape(s[1]), g_R.reshape(s[2]), g_T.reshape(s[3])


def render_train(renderer, rays_o, rays_d, near, far, bt_inv, T_pose_21, verts, Ro, To, index=0, t_rand=None, z_vals=None,
                 repack=True, keep_far_field_setting=False):
    """`NeuSRenderer.render` (utils/renderer.py:190-258) as a differentiable function of the networks' parameters (and of
    bt_inv / T_pose_21 for the hand, Ro / To for the object through `convert_obj_to_local`): the render of a training
    step (exp_runner.py:196-201).  Same return keys as `render`.  With `z_vals` [B,S] the sampling is skipped and
    `render_core` runs at those depths (utils/renderer.py:107-177).  The field is re-packed from the modules' current
    parameters first (`repack=False`: the caller has just done so): a training render follows an optimiser step, and a
    fused step does not advance the version counters `renderer.field()` watches.  `keep_far_field_setting`: leave
    `renderer.compact_far_field` as the caller set it (`NeuSRenderer.render`'s dispatch; the training loop switches the exact
    far-field aggregation on)."""
    from .renderer import _Workspace
    if renderer.perturb <= 0:
        raise ValueError('render requires perturb > 0, as the reference does')
    if not hasattr(renderer, '_ws_train'):
        renderer._ws_train = _Workspace()
    renderer.index = index
    renderer.pack_eval_only = True     # the per-step re-pack builds the evaluation programs only (HN_PACK_EVAL_ONLY)
    if renderer.model_type == 'hand' and getattr(renderer, 'train_compact', True) and not keep_far_field_setting:
        # exact far-field aggregation (hn_field_set_compaction): render and backward pass run on the samples with a live bone mask
        # plus ONE far sample that carries the summed upstream gradients of all the others (they share its all-zero input, so
        # their parameter-gradient contributions are that sum times one Jacobian).  `renderer.train_compact = False`: dense.
        renderer.compact_far_field = True
    if repack:
        renderer.mark_parameters_changed()
    dev = rays_o.device
    if renderer.model_type == 'obj':
        rays_o, rays_d = ObjLocalFn.apply(rays_o, rays_d, torch.as_tensor(Ro, device=dev), torch.as_tensor(To, device=dev))
        bt_inv = T_pose_21 = None
    color, wsum, gerr, cdf, wmax = SingleRenderFn.apply(renderer, near, far, t_rand, z_vals, rays_o, rays_d, bt_inv, T_pose_21,
                                                        *trainable_parameters(renderer))
    B = color.shape[0]
    return {
        'color_fine': color,
        's_val': renderer.field().s_val(B, dev),
        'cdf_fine': cdf,
        'weight_sum': wsum,
        'weight_max': wmax,
        'gradient_error': gerr,
    }


def allreduce_gradients(params, dist=None, average=True):
    """Data-parallel training of one scene over the GPUs of a node (not in the reference, whose `exp_runner.train` is
    single-GPU): every rank renders its own ray batch, the parameter gradients are summed in ONE all-reduce of the
    flattened block (~1.4 M floats for the hand nets: one 5.6 MB ring pass over xGMI per iteration, no per-tensor
    calls) and divided by the world size, so that every rank's Adam takes the same step and the replicas stay
    bit-identical.  `dist`: torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    # every parameter takes part, a missing .grad as zeros: the block has the same size on all ranks whatever each
    # rank's batch touched (a rank-dependent block size would mismatch or hang the collective)
    ps = list(params)
    if not ps:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p in ps:
        n = p.numel()
        g = flat[off:off + n].reshape(p.shape)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def train_step(renderer, optimizer, rays_o, rays_d, near, far, bt_inv, T_pose_21, Ro, To, true_rgb, true_mask, igr_weight=0.1,
               mask_weight=0.1, t_rand=None, extra_loss=None, dist=None):
    """One iteration of exp_runner.train's inner loop (exp_runner.py:196-232): render, loss, backward, optimiser step.
    `extra_loss(render_out)` adds a torch term on the render outputs (the VGG loss of :213-224).  With an initialised
    `torch.distributed` the gradients are averaged over the ranks before the step (`allreduce_gradients`)."""
    out = render_train(renderer, rays_o, rays_d, near, far, bt_inv, T_pose_21, None, Ro, To, t_rand=t_rand)
    terms = train_loss(out, true_rgb, true_mask, igr_weight, mask_weight)
    if extra_loss is not None:
        terms['loss'] = terms['loss'] + extra_loss(out)
    optimizer.zero_grad(set_to_none=True)
    from .fitting import _unit_gradient
    terms['loss'].backward(gradient=_unit_gradient(terms['loss']))
    # everything the optimiser steps (se3_refine and a caller's extra parameters included), so that no replica drifts
    allreduce_gradients([p for g in optimizer.param_groups for p in g['params']], dist)
    optimizer.step()
    renderer.mark_parameters_changed()     # a fused step does not advance the version counters renderer.field() watches
    return terms


FUSED_TRAIN_LOSS = os.environ.get('HONERF_TRAIN_FUSED_LOSS', '1') != '0'    # train_loss on the device as one launch each way (hn_train_loss)


class TrainLossFn(torch.autograd.Function):
    """(color_fine [B,3], weight_sum [B,1], gradient_error []) -> (loss [], terms [6] = loss, colour, mask, eikonal, psnr, mask_sum -- not
    differentiable): exp_runner.py:202-212 as hn_train_loss / hn_train_loss_bwd instead of ~22 + ~25 element-wise torch launches that sit
    between the render's final evaluation and its adjoint (0.2 ms of a 4 ms iteration)."""

    @staticmethod
    def forward(ctx, color_fine, weight_sum, gradient_error, true_rgb, true_mask, igr_weight, mask_weight):
        L = _lib
        lib = L.load()
        dev = color_fine.device
        c, w = L.f32(color_fine).reshape(-1, 3), L.f32(weight_sum).reshape(-1)
        t, m = L.f32(true_rgb, dev).reshape(-1, 3), L.f32(true_mask, dev).reshape(-1)
        ge = L.f32(gradient_error).reshape(1)
        B = c.shape[0]
        assert w.shape[0] == B and t.shape[0] == B and m.shape[0] == B, 'train loss: one colour, weight sum, target and mask per ray'
        terms = torch.empty(6, device=dev, dtype=torch.float32)
        ctx.set_materialize_grads(False)       # (no zero tensor for the gradient of `terms`, which nothing reads)
        L.check(lib.hn_train_loss(L.ptr(c), L.ptr(w), L.ptr(ge), L.ptr(t), L.ptr(m), B, float(igr_weight), float(mask_weight), L.ptr(terms), L.stream_ptr()),
                'hn_train_loss')
        ctx.save_for_backward(c, w, t, m, terms)
        ctx.weights = (float(igr_weight), float(mask_weight))
        ctx.shapes = (color_fine.shape, weight_sum.shape, gradient_error.shape)
        ctx.mark_non_differentiable(terms)
        return terms[0], terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        L = _lib
        lib = L.load()
        if g_loss is None:
            return (None,) * 7
        c, w, t, m, terms = ctx.saved_tensors
        B = c.shape[0]
        gc, gw, gg = torch.empty_like(c), torch.empty_like(w), torch.empty(1, device=c.device, dtype=torch.float32)
        gl = L.f32(g_loss).reshape(1)
        L.check(lib.hn_train_loss_bwd(L.ptr(c), L.ptr(w), L.ptr(t), L.ptr(m), B, L.ptr(terms), L.ptr(gl), ctx.weights[0], ctx.weights[1], L.ptr(gc), L.ptr(gw),
                                      L.ptr(gg), L.stream_ptr()), 'hn_train_loss_bwd')
        s_c, s_w, s_g = ctx.shapes
        return gc.reshape(s_c), gw.reshape(s_w), gg.reshape(s_g), None, None, None, None


def train_loss(render_out, true_rgb, true_mask, igr_weight=0.1, mask_weight=0.1):
    """The loss of exp_runner.py:202-212 without the VGG term (:213-224, a torch module on color_fine).  On the device: one launch forward
    and one backward (TrainLossFn); FUSED_TRAIN_LOSS = False / CPU tensors: the reference's statements as torch operators."""
    color_fine, weight_sum = render_out['color_fine'], render_out['weight_sum']
    if FUSED_TRAIN_LOSS and color_fine.is_cuda and color_fine.dtype == torch.float32 and color_fine.shape[0] > 0:
        loss, terms = TrainLossFn.apply(color_fine, weight_sum, render_out['gradient_error'], true_rgb, true_mask, float(igr_weight), float(mask_weight))
        # (the eikonal term as the caller's own tensor: what it logs is what the render returned)
        return dict(loss=loss, color_fine_loss=terms[1], mask_loss=terms[2], eikonal_loss=render_out['gradient_error'], psnr=terms[4])
    true_mask = (true_mask > 0.5).float()
    mask_sum = true_mask.sum() + 1e-5
    color_error = (color_fine - true_rgb) * true_mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / mask_sum
    psnr = 20.0 * torch.log10(1.0 / (((color_fine - true_rgb) ** 2 * true_mask).sum() / (mask_sum * 3.0)).sqrt())
    mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), true_mask)
    eikonal_loss = render_out['gradient_error']
    loss = color_fine_loss + mask_loss * mask_weight + eikonal_loss * igr_weight
    return dict(loss=loss, color_fine_loss=color_fine_loss, mask_loss=mask_loss, eikonal_loss=eikonal_loss, psnr=psnr.detach())


# ---- the rest of exp_runner's training surface: optimiser, learning-rate schedule, checkpoints (host code) ------------
DEVICE_ADAM = os.environ.get('HONERF_TRAIN_TORCH_ADAM', '0') != '1'    # Adam of the networks as hn_adam_step launches (fitting.PoseAdam)


def make_optimizer(renderer, learning_rate, extra_params=()):
    """Adam over `sdf_network.parameters() + deviation_network.parameters() + color_network.parameters()` in that order
    (exp_runner.py:107-110; `se3_refine` is one of sdf_network's parameters), so that the optimiser state of a
    reference checkpoint loads by index.  On the GPU: `fitting.PoseAdam` -- torch.optim.Adam's defaults and update formula, its
    `param_groups` / `zero_grad` / `step` / `state_dict` (in torch.optim.Adam's layout: checkpoints interchange), the 43 tensors
    in three hn_adam_step launches (~25 us) where torch's fused multi-tensor step takes two of ~43 us each.
    HONERF_TRAIN_TORCH_ADAM=1 (or CPU / non-fp32 parameters): torch.optim.Adam."""
    params = list(renderer.sdf_network.parameters()) + list(renderer.deviation_network.parameters()) + \
        list(renderer.color_network.parameters()) + list(extra_params)
    if DEVICE_ADAM and params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
        from .fitting import PoseAdam
        return PoseAdam([{'params': params, 'lr': learning_rate}])
    try:
        return torch.optim.Adam(params, lr=learning_rate, fused=all(p.is_cuda for p in params))
    except (RuntimeError, TypeError):
        return torch.optim.Adam(params, lr=learning_rate)


def learning_rate_factor(iter_step, warm_up_end, end_iter, learning_rate_alpha):
    """exp_runner.py:266-272: linear warm-up, then cosine decay to `learning_rate_alpha`."""
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - learning_rate_alpha) + learning_rate_alpha


def update_learning_rate(optimizer, iter_step, learning_rate, warm_up_end, end_iter, learning_rate_alpha):
    """exp_runner.py:266-274."""
    f = learning_rate_factor(iter_step, warm_up_end, end_iter, learning_rate_alpha)
    for g in optimizer.param_groups:
        g['lr'] = learning_rate * f
    return learning_rate * f


def save_checkpoint(base_exp_dir, renderer, optimizer, iter_step):
    """exp_runner.py:296-306: same keys, same file name (`checkpoints/ckpt_{iter:06d}.pth`); the reference loads it."""
    ckpt = {
        'sdf_network_fine': renderer.sdf_network.state_dict(),
        'variance_network_fine': renderer.deviation_network.state_dict(),
        'color_network_fine': renderer.color_network.state_dict(),
        'barf_encoding': {},                      # the reference's Embedding has no parameters or buffers
        'optimizer': optimizer.state_dict(),
        'iter_step': int(iter_step),
    }
    d = os.path.join(base_exp_dir, 'checkpoints')
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, 'ckpt_{:0>6d}.pth'.format(int(iter_step)))
    torch.save(ckpt, path)
    return path


def load_checkpoint(path, renderer, optimizer=None, map_location=None):
    """exp_runner.py:288-294 (+ the optimiser state when an optimiser is given) -> iter_step.  The renderer re-packs its
    field on the next call (the parameters' versions change)."""
    ckpt = torch.load(path, map_location=map_location)
    renderer.sdf_network.load_state_dict(ckpt['sdf_network_fine'], strict=False)
    renderer.deviation_network.load_state_dict(ckpt['variance_network_fine'])
    renderer.color_network.load_state_dict(ckpt['color_network_fine'], strict=False)
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    return int(ckpt['iter_step'])


def latest_checkpoint(base_exp_dir):
    """The newest `checkpoints/ckpt_*.pth` (exp_runner.py:112-123 picks the last one by name when `is_continue`)."""
    d = os.path.join(base_exp_dir, 'checkpoints')
    if not os.path.isdir(d):
        return None
    names = sorted(n for n in os.listdir(d) if n.startswith('ckpt_') and n.endswith('.pth'))
    return os.path.join(d, names[-1]) if names else None


def train(renderer, batches, end_iter, base_exp_dir, learning_rate=1e-4, learning_rate_alpha=0.05, warm_up_end=5000,
          igr_weight=1.0, mask_weight=1.0, near=0.4, far=1.5, save_freq=10000, report_freq=100, is_continue=False,
          extra_loss=None, step_fn=None, dist=None):
    """The loop of exp_runner.train (exp_runner.py:126-264) without its dataset side: `batches` yields dicts with
    `rays_o, rays_d [B,3], true_rgb [B,3], true_mask [B,1]` and the frame's `bt_inv, T_pose_21` (hand) or `Ro, To`
    (object) -- what exp_runner.py:136-201 prepares per iteration.  Learning-rate schedule, checkpoints (same keys and
    names, resumable with `is_continue`) and one JSON line of metrics per `report_freq` iterations in
    `<base_exp_dir>/metrics.jsonl` (the scalars the reference sends to TensorBoard, :233-241).  `step_fn` replaces
    `train_step` in host-only tests."""
    import json
    step_fn = step_fn or train_step
    optimizer = make_optimizer(renderer, learning_rate)
    iter_step = 0
    if is_continue:
        ck = latest_checkpoint(base_exp_dir)
        if ck is not None:
            iter_step = load_checkpoint(ck, renderer)      # networks and iteration count; Adam's moments restart, as in the reference (:288-293)
    # the process group as train_step -> allreduce_gradients resolves it (an initialised torch.distributed counts whether or
    # not the caller passed the module), so that logging, checkpoints and the barrier agree with what the gradients do
    if dist is None:
        import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank0 = not multi or dist.get_rank() == 0
    os.makedirs(base_exp_dir, exist_ok=True)
    log_path = os.path.join(base_exp_dir, 'metrics.jsonl')
    update_learning_rate(optimizer, iter_step, learning_rate, warm_up_end, end_iter, learning_rate_alpha)
    it = iter(batches)
    while iter_step < end_iter:
        try:
            b = next(it)
        except StopIteration:          # one epoch of the dataloader is over: start the next (exp_runner.py:133-134)
            it = iter(batches)
            b = next(it)
        terms = step_fn(renderer, optimizer, b['rays_o'], b['rays_d'], near, far, b.get('bt_inv'), b.get('T_pose_21'), b.get('Ro'),
                        b.get('To'), b['true_rgb'], b['true_mask'], igr_weight, mask_weight, extra_loss=extra_loss, dist=dist)
        iter_step += 1
        if iter_step % report_freq == 0:
            rec = {'iter': iter_step, 'lr': optimizer.param_groups[0]['lr']}
            rec.update({k: float(v.detach()) if isinstance(v, torch.Tensor) else float(v) for k, v in terms.items()})
            if rank0:
                with open(log_path, 'a') as f:
                    f.write(json.dumps(rec) + '\n')
        if iter_step % save_freq == 0:
            if rank0:                  # replicas are identical: one writer, and nobody runs ahead of a half-written file
                save_checkpoint(base_exp_dir, renderer, optimizer, iter_step)
            if multi:
                dist.barrier()
        update_learning_rate(optimizer, iter_step, learning_rate, warm_up_end, end_iter, learning_rate_alpha)
    renderer.pack_eval_only = False        # leaving the training path: the next pack builds every program again
    if hasattr(renderer, 'mark_parameters_changed'):
        renderer.mark_parameters_changed()
    return iter_step

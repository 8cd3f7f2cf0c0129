"""The hand pose chain of the fitting loops as one differentiable op.

Per optimisation step the reference turns its refine parameters into the hand field's `bone_transformation_inv` with
(fitting_single.py:206-226, the same statements in fitting_video.py)

    convert_joints -> transform_to_canonical -> PoseConverter.get_refine_3d_joint -> inverse canonical transform ->
    rot6d_to_matrix / palm rotation + translation -> convert_joints -> transform_to_canonical -> PoseConverter.forward ->
    convert_joints -> @ rot_then_swap_mat

which is about 4 000 small torch operators forward (halo_util/converter_fit_batch.py) and as many backward: with the
renderer at a few milliseconds per step it is the step's largest part.  `hn_pose_chain` evaluates the whole chain in one
launch together with its exact Jacobian (forward-mode dual numbers, ho-nerf_amd/csrc/hn_pose_chain.h); the backward pass is
one more launch (`hn_pose_chain_bwd`).  Parity: tests/golden/pose_chain.npz holds values and Jacobians produced by
executing the reference's own statements (tests/golden/make_golden.py, pose_goldens).
"""
import torch

from . import lib as _lib

N_IN, N_OUT = 36, 399


class PoseChainFn(torch.autograd.Function):
    """(ori_pose [F,21,3], bone_len [F,20], params [F,36]) -> (bone_transformation_inv [F,21,4,4], joint_3d [F,21,3]);
    params = [joint_refine_angle 20 | palm_refine_angle 7 | palm_rot_refine 6 | palm_trans_refine 3].  Gradients flow to
    `params` only: the predicted joints and bone lengths are data in the reference's loop too."""

    @staticmethod
    def forward(ctx, ori_pose, bone_len, params):
        L = _lib
        lib = L.load()
        ori, bl, prm = L.f32(ori_pose).reshape(-1, 21, 3), L.f32(bone_len).reshape(-1, 20), L.f32(params).reshape(-1, N_IN)
        F, dev = ori.shape[0], ori.device
        assert bl.shape[0] == F and prm.shape[0] == F, 'one row of bone lengths and parameters per frame'
        bt = torch.empty(F, 21, 4, 4, device=dev, dtype=torch.float32)
        j3 = torch.empty(F, 21, 3, device=dev, dtype=torch.float32)
        need = params.requires_grad
        jac = torch.empty(F, N_OUT, N_IN, device=dev, dtype=torch.float32) if need else None
        L.check(lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, L.ptr(bt), L.ptr(j3), L.ptr(jac) if need else None,
                                  L.stream_ptr()), 'hn_pose_chain')
        ctx.jac, ctx.shape = jac, params.shape
        return bt, j3

    @staticmethod
    def backward(ctx, g_bt, g_j3):
        L = _lib
        lib = L.load()
        jac = ctx.jac
        F = jac.shape[0]
        g = torch.empty(F, N_IN, device=jac.device, dtype=torch.float32)
        z = lambda t, n: torch.zeros(F, n, device=jac.device) if t is None else L.f32(t).reshape(F, n)
        go = torch.cat([z(g_bt, 336), z(g_j3, 63)], dim=1)
        L.check(lib.hn_jacobian_vjp(L.ptr(jac), L.ptr(go), F, N_OUT, N_IN, L.ptr(g), L.stream_ptr()), 'hn_jacobian_vjp')
        return None, None, g.reshape(ctx.shape)


def pose_chain(ori_3d_pose, cur_bone_length, joint_refine_angle, palm_refine_angle, palm_rot_refine, palm_trans_refine):
    """fitting_single.py:206-226 with the reference's variable names: ori_3d_pose [F,21,3] (MANO order), cur_bone_length
    [F,20], joint_refine_angle [F,20], palm_refine_angle [F,7], palm_rot_refine [F,3,2], palm_trans_refine [F,3] ->
    (bone_transformation_inv [F,21,4,4], joint_3d [F,21,3])."""
    F = ori_3d_pose.shape[0]
    params = torch.cat([joint_refine_angle.reshape(F, 20), palm_refine_angle.reshape(F, 7), palm_rot_refine.reshape(F, 6),
                        palm_trans_refine.reshape(F, 3)], dim=1)
    return PoseChainFn.apply(ori_3d_pose, cur_bone_length, params)


class HandPoseChain(torch.nn.Module):
    """The hand half of the fitting loops' parameter set (fitting_single.py:183-198) over `hn_pose_chain`: the four refine
    leaves as one [F,36] parameter (views with the reference's names), forward() -> (bone_transformation_inv, joint_3d)."""

    def __init__(self, ori_3d_pose, cur_bone_length):
        super().__init__()
        ori = torch.as_tensor(ori_3d_pose, dtype=torch.float32).reshape(-1, 21, 3)
        F = ori.shape[0]
        self.register_buffer('ori_3d_pose', ori.clone())
        self.register_buffer('cur_bone_length', torch.as_tensor(cur_bone_length, dtype=torch.float32).reshape(F, 20).clone())
        init = torch.zeros(F, N_IN)
        init[:, 27:33] = torch.eye(3)[:, :2].reshape(-1)   # palm_rot_refine = eye(3)[:, :2] (fitting_single.py:183-185)
        self.params = torch.nn.Parameter(init)

    joint_refine_angle = property(lambda self: self.params[:, 0:20])
    palm_refine_angle = property(lambda self: self.params[:, 20:27])
    palm_rot_refine = property(lambda self: self.params[:, 27:33].reshape(-1, 3, 2))
    palm_trans_refine = property(lambda self: self.params[:, 33:36])

    def forward(self):
        return PoseChainFn.apply(self.ori_3d_pose, self.cur_bone_length, self.params)


class RigidPoseFn(torch.autograd.Function):
    """hn_rigid_pose: params [F,18] = [obj_rot 6 | obj_trans 3 | palm_rot 6 | palm_trans 3] -> out [F,412] =
    [bt_inv 336 | joint_3d 63 | obj_r 9 | obj_t 3 | joint loss 1] (fitting_single.py:213-217, 227-231, 260); with_palm False:
    the object half only (the hand half then comes from PoseChainFn).  Backward = hn_jacobian_vjp on the stored Jacobian."""

    @staticmethod
    def forward(ctx, params, bt_inv0, joints0, Ro_pred, To_pred, with_palm):
        L = _lib
        lib = L.load()
        prm = L.f32(params).reshape(-1, 18)
        F, dev = prm.shape[0], prm.device
        out = torch.zeros(F, 412, device=dev, dtype=torch.float32) if not with_palm else torch.empty(F, 412, device=dev, dtype=torch.float32)
        need = params.requires_grad
        jac = (torch.zeros if not with_palm else torch.empty)(F, 412, 18, device=dev, dtype=torch.float32) if need else None
        L.check(lib.hn_rigid_pose(L.ptr(bt_inv0) if with_palm else None, L.ptr(joints0) if with_palm else None, L.ptr(Ro_pred), L.ptr(To_pred),
                                  L.ptr(prm), F, 1 if with_palm else 0, L.ptr(out), L.ptr(jac) if need else None, L.stream_ptr()), 'hn_rigid_pose')
        ctx.jac, ctx.shape = jac, params.shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib
        lib = L.load()
        jac = ctx.jac
        F = jac.shape[0]
        g = torch.empty(F, 18, device=jac.device, dtype=torch.float32)
        go = L.f32(g_out).reshape(F, 412)
        L.check(lib.hn_jacobian_vjp(L.ptr(jac), L.ptr(go), F, 412, 18, L.ptr(g), L.stream_ptr()), 'hn_jacobian_vjp')
        return g.reshape(ctx.shape), None, None, None, None, None


class VertsLossFn(torch.autograd.Function):
    """pose_loss between the vertex sets of two rigid poses (fitting_single.py:232-233, fitting_video.py's vertex smoothness):
    (Ra [P,3,3], ta [P,3], Rb, tb, verts [V,3]) -> loss [P] = mean_v |(Ra - Rb) v + (ta - tb)|, one launch with the
    closed-form gradient (hn_verts_loss); the vertex sets themselves are never formed."""

    @staticmethod
    def forward(ctx, Ra, ta, Rb, tb, verts):
        L = _lib
        lib = L.load()
        a, b, c, d = L.f32(Ra).reshape(-1, 9), L.f32(ta).reshape(-1, 3), L.f32(Rb).reshape(-1, 9), L.f32(tb).reshape(-1, 3)
        P, dev = a.shape[0], a.device
        loss = torch.empty(P, device=dev, dtype=torch.float32)
        gR, gt = torch.empty(P, 9, device=dev, dtype=torch.float32), torch.empty(P, 3, device=dev, dtype=torch.float32)
        L.check(lib.hn_verts_loss(L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(d), L.ptr(verts), verts.shape[0], P, L.ptr(loss), L.ptr(gR), L.ptr(gt),
                                  L.stream_ptr()), 'hn_verts_loss')
        ctx.save_for_backward(gR, gt)
        ctx.shapes = (Ra.shape, ta.shape, Rb.shape, tb.shape)
        ctx.needs = (Ra.requires_grad, ta.requires_grad, Rb.requires_grad, tb.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        gR, gt = ctx.saved_tensors
        g = g_loss.reshape(-1, 1)
        dR, dt = gR * g, gt * g
        sh, nd = ctx.shapes, ctx.needs
        return (dR.reshape(sh[0]) if nd[0] else None, dt.reshape(sh[1]) if nd[1] else None,
                (-dR).reshape(sh[2]) if nd[2] else None, (-dt).reshape(sh[3]) if nd[3] else None, None)


_ZEROS9 = {}


def _zeros9(F, dev):
    """A cached [F, 9] zero block (the unused inputs of hn_rigid_pose); never written."""
    key = (F, str(dev))
    if key not in _ZEROS9:
        _ZEROS9[key] = torch.zeros(F, 9, device=dev, dtype=torch.float32)
    return _ZEROS9[key]


_AUX_STREAMS = {}
_DEFER_JACOBIAN = [False]
_DEFERRED = []


class deferred_jacobians:
    """with deferred_jacobians(): ... -- HaloChainFn.forward inside the block does not launch the chain's Jacobian on the extra stream
    but leaves it pending; `flush_deferred_jacobians()` launches what is pending (the backward pass would, at the latest).  For
    callers that put their own work on that stream first."""

    def __enter__(self):
        self.prev = _DEFER_JACOBIAN[0]
        _DEFER_JACOBIAN[0] = True

    def __exit__(self, *exc):
        _DEFER_JACOBIAN[0] = self.prev


def flush_deferred_jacobians():
    while _DEFERRED:
        ctx = _DEFERRED.pop(0)
        if getattr(ctx, 'jac_pending', None) is not None:
            ctx.jac_pending()


JACOBIAN_ON_AUX = True     # HaloChainFn: the chain's Jacobian launch on the extra stream beside the render's sampling (False: on the caller's stream)


def _aux_stream(dev):
    """The stream the pose chain's Jacobian launches run on (one per device)."""
    key = str(dev)
    if key not in _AUX_STREAMS:
        _AUX_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _AUX_STREAMS[key]


_BOUND = set()


def bind_streams(dev):
    """Touch the library's second stream and the extra stream of `dev` NOW, so that the runtime gives them hardware queues of
    their own before other streams of the process (a communicator's, a data loader's) ask for theirs.  The runtime multiplexes
    streams onto a few hardware queues (4 by default) in the order of their first use, a later stream shares a queue with an
    earlier one, and it matters which: a mostly idle stream sharing the caller's queue costs nothing, the object branch's 1.2 ms
    adjoint kernel in front of the caller's launches costs a millisecond per fitting step (`tools/hw_queue_probe.py`: 2.54 ms with
    these streams bound first whatever comes later, 3.7 ms when two other streams came first)."""
    import ctypes
    dev = torch.device(dev)
    if dev.type != 'cuda' or str(dev) in _BOUND:
        return
    _BOUND.add(str(dev))
    L = _lib
    lib = L.load()
    with torch.cuda.device(dev):
        sp = ctypes.c_void_p()
        L.check(lib.hn_side_stream(ctypes.byref(sp)), 'hn_side_stream')
        side = torch.cuda.ExternalStream(sp.value, device=dev)
        for st in (side, _aux_stream(dev)):
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev)
            torch.cuda.current_stream(dev).wait_stream(st)


class HaloChainFn(torch.autograd.Function):
    """The whole pose side of a fitting_single step as ONE autograd node over the six refine leaves (fitting_single.py:177-235):
    (obj_rot [F,3,2], obj_trans [F,3], palm_rot [F,3,2], palm_trans [F,3], joint_refine_angle [F,20], palm_refine_angle [F,7])
    -> (bone_transformation_inv [F,21,4,4], joint_3d [F,21,3], obj_r [F,3,3], obj_t [F,3]) -- hn_pose_chain + hn_rigid_pose forward,
    two hn_jacobian_vjp backward.  As separate nodes joined by cat / slice operators the same graph was ~12 small launches
    forward and ~25 backward (every slice's backward is a zero fill and a copy)."""

    @staticmethod
    def forward(ctx, obj_rot, obj_trans, palm_rot, palm_trans, joint_angle, palm_angle, ori_pose, bone_len, Ro_pred, To_pred, rows=None):
        """rows (int64 [F] or None): the leaves hold n >= F frames and the chain runs on these rows of them (a fitting_video
        window; ori_pose .. To_pred are already the window's): one gather forward, one scatter backward, instead of an advanced
        index per leaf whose backward is a sort-based index_put each."""
        L = _lib
        lib = L.load()
        F, dev = ori_pose.shape[0], ori_pose.device
        st = L.stream_ptr()
        n = obj_rot.shape[0]
        if rows is not None:
            # the kernels read the index as int64 whatever the caller's tensor held (an int32 / uint8 index tensor was accepted by
            # index_select's successors here and read as garbage); rows outside [0, n) are never dereferenced (hn_leaf_rows_gather)
            if rows.dtype == torch.bool or rows.dim() != 1 or rows.shape[0] != F:
                raise IndexError('HaloChainFn: rows must be one integer frame id per frame of the window')
            rows = rows.to(device=dev, dtype=torch.long).contiguous()
        ctx.rows, ctx.n = rows, n
        leaves = (obj_rot, obj_trans, palm_rot, palm_trans, joint_angle, palm_angle)
        if rows is not None and all(x.is_contiguous() and x.dtype == torch.float32 for x in leaves):
            # a window of a sequence: the rows of the six leaves -> the two input blocks, one launch (hn_leaf_rows_gather)
            import ctypes
            prm_h = torch.empty(F, 36, device=dev, dtype=torch.float32)
            prm_o = torch.empty(F, 18, device=dev, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * 6)(*[x.data_ptr() for x in leaves])
            L.check(lib.hn_leaf_rows_gather(ptrs, L.ptr(rows), F, n, L.ptr(prm_h), L.ptr(prm_o), st), 'hn_leaf_rows_gather')
        else:
            prm = torch.cat([joint_angle.reshape(n, 20), palm_angle.reshape(n, 7), palm_rot.reshape(n, 6), palm_trans.reshape(n, 3),
                             obj_rot.reshape(n, 6), obj_trans.reshape(n, 3), _zeros9(n, dev)], dim=1)    # [n, 45 + 9]: one launch
            if rows is not None:
                prm = prm.index_select(0, rows)
            prm_h = prm[:, :36].contiguous() if F > 1 else prm[:, :36]
            prm_o = prm[:, 36:54].contiguous() if F > 1 else prm[:, 36:54]                           # hn_rigid_pose's 18 inputs: 9 used here
        need = any(x.requires_grad for x in (obj_rot, obj_trans, palm_rot, palm_trans, joint_angle, palm_angle))
        bt = torch.empty(F, 21, 4, 4, device=dev, dtype=torch.float32)
        j3 = torch.empty(F, 21, 3, device=dev, dtype=torch.float32)
        jac_h = torch.empty(F, N_OUT, N_IN, device=dev, dtype=torch.float32) if need else None
        # values and Jacobian as two launches: what follows (the render) waits for the values, a third of the chain's time; the
        # Jacobian is read by backward() and runs on a stream of its own beside the render's sampling
        ctx.jac_ev = None
        ctx.jac_pending = None
        if need and not JACOBIAN_ON_AUX:
            L.check(lib.hn_pose_chain(L.ptr(ori_pose), L.ptr(bone_len), None, L.ptr(prm_h), F, None, None, L.ptr(jac_h), st), 'hn_pose_chain')
        elif need:
            aux = _aux_stream(dev)
            ready = torch.cuda.Event()
            ready.record()

            def launch_jacobian():
                aux.wait_event(ready)
                L.check(lib.hn_pose_chain(L.ptr(ori_pose), L.ptr(bone_len), None, L.ptr(prm_h), F, None, None, L.ptr(jac_h), aux.cuda_stream), 'hn_pose_chain')
                ctx.jac_ev = torch.cuda.Event()
                ctx.jac_ev.record(aux)
                ctx.jac_pending = None
            for t in (prm_h, jac_h, ori_pose, bone_len):
                t.record_stream(aux)
            if _DEFER_JACOBIAN[0]:
                # the caller has more work for the extra stream that the step needs SOONER than the Jacobian (fitting_video's stable
                # term, whose value the loss waits for): it launches the Jacobian behind that work (flush_deferred_jacobians)
                ctx.jac_pending = launch_jacobian
                _DEFERRED.append(ctx)
            else:
                launch_jacobian()
        L.check(lib.hn_pose_chain(L.ptr(ori_pose), L.ptr(bone_len), None, L.ptr(prm_h), F, L.ptr(bt), L.ptr(j3), None, st), 'hn_pose_chain')
        # (object half only: entries 399 .. 410 of out / jac_o are written, and only those are read below)
        out = torch.empty(F, 412, device=dev, dtype=torch.float32)
        jac_o = torch.empty(F, 412, 18, device=dev, dtype=torch.float32) if need else None
        L.check(lib.hn_rigid_pose(None, None, L.ptr(Ro_pred), L.ptr(To_pred), L.ptr(prm_o), F, 0, L.ptr(out), L.ptr(jac_o) if need else None, st),
                'hn_rigid_pose')
        ctx.jac_h, ctx.jac_o, ctx.F = jac_h, jac_o, F
        ctx.shapes = tuple(x.shape for x in (obj_rot, obj_trans, palm_rot, palm_trans, joint_angle, palm_angle))
        # (F > 1: the two slices of the [F, 412] block are strided; every consumer -- the loss node, the inverse, the stable term, the
        #  render -- would make its own contiguous copy: one copy each here instead)
        return bt, j3, out[:, 399:408].contiguous().reshape(F, 3, 3), out[:, 408:411].contiguous()

    @staticmethod
    def backward(ctx, g_bt, g_j3, g_or, g_ot):
        L = _lib
        lib = L.load()
        F = ctx.F
        dev = ctx.jac_h.device
        from .autograd import _join_pending_side
        _join_pending_side(dev)      # (upstream gradients produced on a side stream by the loss node: joined by whoever consumes them first)
        if ctx.jac_pending is not None:      # (deferred and never flushed: now)
            ctx.jac_pending()
        c = lambda t, n: None if t is None else L.f32(t).reshape(F, n)
        gb, gj, gr, gt = c(g_bt, 336), c(g_j3, 63), c(g_or, 9), c(g_ot, 3)
        g = torch.empty(F, 45, device=dev, dtype=torch.float32)
        if ctx.jac_ev is not None:
            torch.cuda.current_stream().wait_event(ctx.jac_ev)
        # both Jacobian products in one launch (hn_pose_side_vjp); a missing upstream gradient counts as zero there
        L.check(lib.hn_pose_side_vjp(L.ptr(ctx.jac_h), L.ptr(ctx.jac_o), L.ptr(gb), L.ptr(gj), L.ptr(gr), L.ptr(gt), None, None, F, 3, L.ptr(g),
                                     L.stream_ptr()),
                'hn_pose_side_vjp')
        gh, go = g[:, :36], g[:, 36:45]
        sh = ctx.shapes
        if ctx.rows is not None:
            # the window's rows of six contiguous gradient blocks (zero elsewhere): one fill and one launch (hn_leaf_rows_scatter);
            # contiguous, so autograd keeps them as the leaves' .grad without a copy each
            n = ctx.n
            out = torch.zeros(n * 45, device=dev, dtype=torch.float32)
            L.check(lib.hn_leaf_rows_scatter(L.ptr(g), L.ptr(ctx.rows), F, n, L.ptr(out), L.stream_ptr()), 'hn_leaf_rows_scatter')
            return (out[:6 * n].view(sh[0]), out[6 * n:9 * n].view(sh[1]), out[9 * n:15 * n].view(sh[2]), out[15 * n:18 * n].view(sh[3]),
                    out[18 * n:38 * n].view(sh[4]), out[38 * n:45 * n].view(sh[5]), None, None, None, None, None)
        # views of one block (for F = 1 every slice is contiguous: autograd keeps them as the leaves' .grad without a copy)
        return (go[:, 0:6].reshape(sh[0]), go[:, 6:9].reshape(sh[1]), gh[:, 27:33].reshape(sh[2]), gh[:, 33:36].reshape(sh[3]),
                gh[:, 0:20].reshape(sh[4]), gh[:, 20:27].reshape(sh[5]), None, None, None, None, None)

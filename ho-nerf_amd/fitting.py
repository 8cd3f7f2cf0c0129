"""Frame-sharded host logic of the fitting loops (fitting_single.py / fitting_video.py).

What is here is the part of the callers that decides WHICH frames / windows a rank works on, the
render-dependent loss terms they evaluate on the renderer's outputs, and the one reduction the
multi-GPU layout needs.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests); frames of `fitting_single` are independent optimisation problems
(own parameters and Adam state, fitting_single.py:143, 177-199), so they shard with NO data-path
collective -- the only exchange is the sum of a small loss/metric vector per logging interval.

`fitting_video` updates shared pose parameters window by window (fitting_video.py:146-149, 340-342);
`window_schedule` gives the synchronous window-parallel assignment of SURVEY 8(e) (rank r takes
window step*world + r).  A step is split into `fit_backward` (pose chain -> render -> losses -> backward into
the pose parameters) and `fit_apply` (one SUM all-reduce of the flattened pose-gradient block, then Adam), so
that the collective sits between the backward pass and the optimiser; `fit_sequence_video` is the sequence loop
over them (5 outer iterations x windows x 4 sub-iterations x views) and `fit_frames_sharded` the per-frame loop
of fitting_single with its skip-if-done restart rule.  With one rank both are the reference's sequential loops.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

LOSS_KEYS = ('loss', 'color', 'mask', 'contact', 'penetration', 'frames')


def shard_frames(n_frames, rank, world):
    """Frames of this rank: strided assignment (frame f -> rank f % world), so that a growing
    sequence keeps every rank busy and a restart with another world size re-shards trivially.
    (FrameShardedRunner deals out the frames that are still to do, i.e. applies this to their positions.)"""
    if not (0 <= rank < world):
        raise ValueError('rank %d outside world of %d' % (rank, world))
    return list(range(rank, n_frames, world))


def sliding_windows(n_frames, window=4):
    """RayImageSampler (utils/dataset.py:384-407): windows [i, i+1, .., i+window-1], i < n-window+1."""
    return [list(range(i, i + window)) for i in range(max(n_frames - window + 1, 0))]


def window_schedule(n_frames, rank, world, window=4):
    """Synchronous window-parallel schedule: at step s rank r owns window s*world + r (None when
    the sequence has run out: the rank then contributes a zero gradient to that step's all-reduce)."""
    wins = sliding_windows(n_frames, window)
    steps = (len(wins) + world - 1) // world
    return [wins[s * world + rank] if s * world + rank < len(wins) else None for s in range(steps)]


def render_loss_terms(render_out, true_rgb, true_mask, fit_type='1', video=False):
    """The render-dependent loss terms of one optimisation step (fitting_single.py:251-283; video=True:
    fitting_video.py:285-309, tensors [F,P,.]).

    color: L1(sum) of the masked colour error / B (video: / F / P); mask: BCE(clip(weight_sum, 1e-3, 1-1e-3), mask),
    weighted 0.5 (video: the render loss is halved again, fitting_video.py:291); fit_type '12' (and every video
    type) adds contact (mean |s_h| + |s_o| where < 1e-2, x30) and penetration (mean over s_o < 0 and s_h < 0, x20).
    The pose-regularisation terms (joint / object-vertex losses) depend on the pose chain only: step_loss adds them."""
    color_fine = render_out['color_fine']
    weight_sum = render_out['weight_sum']
    interaction = video or fit_type in ('12', '123', '1234')
    if color_fine.is_cuda:
        # on the device the same four terms come from two launches (hn_fit_loss_sums / _grads) instead of ~85
        from .autograd import FitLossFn
        color_loss, mask_loss, contact, penet = FitLossFn.apply(color_fine, weight_sum, render_out['sdf_hand'] if interaction else None,
                                                                render_out['sdf_obj'] if interaction else None, true_rgb, true_mask)
        render_loss = color_loss + 0.5 * mask_loss
        if video:
            render_loss = 0.5 * render_loss
        terms = {'color': color_loss, 'mask': mask_loss, 'loss': render_loss, 'contact': contact, 'penetration': penet}
        if interaction:
            terms['loss'] = terms['loss'] + 30 * contact + 20 * penet
        return terms
    color_error = (color_fine - true_rgb) * true_mask
    color_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / true_mask.shape[0]
    if video:
        color_loss = color_loss / true_mask.shape[1]
    mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), true_mask)
    render_loss = color_loss + 0.5 * mask_loss
    if video:
        render_loss = 0.5 * render_loss
    terms = {'color': color_loss, 'mask': mask_loss, 'loss': render_loss}
    zero = color_loss.new_zeros(())
    terms['contact'], terms['penetration'] = zero, zero
    if interaction:
        # Same sums as fitting_single.py:268-281, written with masks instead of boolean indexing: indexing makes
        # tensors of data-dependent size, i.e. a device -> host synchronisation in the middle of every step
        sdf_hand = render_out['sdf_hand'][:, 0]
        sdf_obj = render_out['sdf_obj'][:, 0]
        sdf_abs_sum = sdf_hand.abs() + sdf_obj.abs()
        contact_id = (sdf_abs_sum < 1e-2).to(sdf_abs_sum.dtype)
        contact = (sdf_abs_sum * contact_id).sum() / (contact_id.sum() + 1e-9)
        pen_id = ((sdf_obj < 0) & (sdf_hand < 0)).to(sdf_abs_sum.dtype)
        penet = (sdf_abs_sum * pen_id).sum() / (pen_id.sum() + 1e-9)
        terms['contact'], terms['penetration'] = contact, penet
        terms['loss'] = terms['loss'] + 30 * contact + 20 * penet
    return terms


class FrameShardedRunner:
    """Runs `frame_fn(frame_id) -> dict of scalar tensors/floats` over this rank's frames and
    reduces the sums over all ranks.

    * restartable like the reference (fitting_single.py:156-158 skips frames whose result file
      exists): pass `done(frame_id) -> bool`;
    * the reduction is one all-reduce of a len(LOSS_KEYS) vector (SUM) -- with one rank it is the
      identity, so 1-GPU and N-GPU runs give the same totals up to summation order.
    """

    def __init__(self, n_frames, rank=None, world=None, device=None, done=None):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = rank if rank is not None else (self.dist.get_rank() if self.dist else int(os.environ.get('RANK', 0)))
        self.world = world if world is not None else (self.dist.get_world_size() if self.dist else 1)
        if device is None:      # RCCL ("nccl") reduces device tensors only; gloo takes host tensors
            on_gpu = self.dist is not None and self.dist.get_backend() == 'nccl'
            device = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
        self.device = device
        # what is left to do is decided BEFORE the frames are dealt out, so that a restart (or a re-sharded run) spreads the
        # missing frames over all ranks instead of leaving the ranks whose frames exist idle; rank 0's view of `done` is the
        # one every rank uses (a frame finishing between two ranks' directory scans must not be dealt twice or not at all)
        todo = torch.tensor([0 if (done and done(f)) else 1 for f in range(n_frames)], dtype=torch.int32, device=device)
        if self.dist is not None and self.world > 1 and n_frames > 0:
            self.dist.broadcast(todo, src=0)
        left = [f for f, t in enumerate(todo.tolist()) if t]
        self.frames = [left[i] for i in shard_frames(len(left), self.rank, self.world)]
        self.totals = torch.zeros(len(LOSS_KEYS), dtype=torch.float64)

    def run(self, frame_fn):
        for f in self.frames:
            terms = frame_fn(f)
            for k, key in enumerate(LOSS_KEYS[:-1]):
                if key in terms:
                    v = terms[key]
                    self.totals[k] += float(v.detach()) if isinstance(v, torch.Tensor) else float(v)
            self.totals[-1] += 1
        return self.reduce()

    def run_groups(self, group_fn, batch):
        """As `run`, the rank's frames taken `batch` at a time: group_fn([frame ids]) -> [terms per frame]."""
        for a in range(0, len(self.frames), batch):
            fs = self.frames[a:a + batch]
            for terms in group_fn(fs):
                for k, key in enumerate(LOSS_KEYS[:-1]):
                    if key in terms:
                        v = terms[key]
                        self.totals[k] += float(v.detach()) if isinstance(v, torch.Tensor) else float(v)
                self.totals[-1] += 1
        return self.reduce()

    def reduce(self):
        t = self.totals.to(self.device)
        self.allreduce_calls = getattr(self, 'allreduce_calls', 0)
        if self.dist is not None and (self.world > 1 or FORCE_COLLECTIVE):
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            self.allreduce_calls += 1
        t = t.cpu()
        n = max(float(t[-1]), 1.0)
        out = {k: float(t[i]) / n for i, k in enumerate(LOSS_KEYS[:-1])}
        out['frames'] = int(t[-1])
        return out


def mask_pixels(mask, n_rays, rng):
    """get_rays_xy (utils/dataset.py:23-50) with threshold 1.0: n_rays random pixels inside the mask ->
    NDC xy = -((px - W/2) / (H/2), (py - H/2) / (H/2)) (:45-47) and the flat pixel indices."""
    H, W = mask.shape
    ys, xs = np.nonzero(mask > 0)
    sel = rng.integers(0, len(ys), size=n_rays)
    py, px = ys[sel], xs[sel]
    x = -(px - W / 2.0) / (H / 2.0)
    y = -(py - H / 2.0) / (H / 2.0)
    return np.stack([x, y], -1).astype(np.float32), py * W + px


# HONERF_FORCE_COLLECTIVE=1 (or fitting.FORCE_COLLECTIVE = True): the collectives of the sharded loops are issued on a process group
# of ONE rank too, where they are the identity -- so that the RCCL path (communicator, device-side all-reduce between the backward
# pass and Adam, the loss reduction of the frame-sharded runner) can be executed and checked on a single GPU.
FORCE_COLLECTIVE = os.environ.get('HONERF_FORCE_COLLECTIVE') == '1'


def _grad_block(params):
    """The gradients of `params` as ONE flat tensor when they already are one: contiguous fp32 views that tile a range of a single
    storage without gaps IN THE ORDER OF `params` (HaloChainFn.backward hands out the six leaves' gradients of a window as views of
    one [n x 45] block laid out in that order), so that the block is element for element what `cat` of the gradients would be --
    every rank must present the same layout to the collective, and a rank without a window presents the cat of zeros.  None otherwise."""
    gs = [p.grad for p in params]
    if not gs or any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in gs):
        return None
    base = gs[0].untyped_storage().data_ptr()
    if any(g.untyped_storage().data_ptr() != base or g.device != gs[0].device for g in gs):
        return None
    spans = [(g.storage_offset(), g.numel()) for g in gs]
    at = spans[0][0]
    for off, n in spans:
        if off != at:
            return None
        at += n
    flat = gs[0].new_empty(0)
    flat.set_(gs[0].untyped_storage(), spans[0][0], (at - spans[0][0],), (1,))
    return flat


def allreduce_pose_gradients(params, dist=None, average=False, force=None):
    """Window-parallel `fitting_video` step: every rank has back-propagated ITS window's loss into the
    shared `[data_num, ...]` pose parameters (non-zero on the window's 4 rows); one all-reduce (SUM) of the flattened
    gradient block -- data_num x 45 floats, ~18 KB at 100 frames: latency-bound, a single call -- makes the gradients
    identical on all ranks, after which every rank takes the same Adam step on its replica.  A rank whose window list
    has run out passes parameters without `.grad` (treated as zeros).  Returns the number of floats exchanged.

    The reduced gradient is that of the SUM of the concurrent windows' losses: a frame covered by several of them
    receives several contributions in one step, where the reference's sequential schedule would take as many Adam
    steps (fitting_video.py:340-342).  That is the schedule change SURVEY 8(e) accepts (Jacobi over `world` windows);
    Adam's per-element normalisation keeps the step size independent of how many windows touched a row.  Pass
    `average=True` to divide by the world size instead (plain data-parallel mean).

    The gradients the device loops produce are views of one contiguous block already (`_grad_block`): that block is reduced IN
    PLACE -- one collective, no gather / scatter launches around it; anything else (a rank without a window, gradients from
    separate allocations) is flattened, reduced and copied back.  `force` (default: FORCE_COLLECTIVE) issues the collective on
    a one-rank group too."""
    if dist is None:
        import torch.distributed as dist
    force = FORCE_COLLECTIVE if force is None else force
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return 0
    params = list(params)
    flat = _grad_block(params)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= dist.get_world_size()
        return int(flat.numel())
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].reshape(p.shape)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return int(flat.numel())


# ---- pose parameters -> render inputs ---------------------------------------------------------------------------------
def rot6d_to_matrix(rot_6d):
    """utils/utils.py:11-29 (Zhou et al. 2019): [...,3,2] -> [B,3,3] with columns (b1, b2, b1 x b2)."""
    r = rot_6d.reshape(-1, 3, 2)
    a1, a2 = r[:, :, 0], r[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def pose_loss(target, pred, mean=False):
    """fitting_single.py:119-122 (sum / rows) and fitting_video.py:123-126 (mean=True: mean over everything)."""
    err = torch.norm(target - pred, dim=-1)
    return err.mean() if mean else err.sum() / err.shape[0]


def _index_tensor(owner, index, device):
    """The window's frame ids as a device tensor, cached on the chain: a window is visited sub_iters x views times per
    pass, and a fresh host list would be a host -> device copy in front of every step."""
    if isinstance(index, torch.Tensor):
        if index.dtype == torch.bool or index.is_floating_point():
            raise IndexError('a window is a list of integer frame ids')
        return index.to(device=device, dtype=torch.long).contiguous()
    cache = owner.__dict__.setdefault('_idx_cache', {})
    key = tuple(int(i) for i in index)
    if key not in cache:
        if len(cache) > 4096:
            cache.clear()
        cache[key] = torch.tensor(key, dtype=torch.long, device=device)
    return cache[key]


class RigidPoseChain:
    """Host-side differentiable map from the optimised parameters to the renderer's pose inputs.

    In the reference this is halo_util's PoseConverter chain (fitting_single.py:206-230): joint angles ->
    biomechanical skeleton -> per-bone world->T-pose transforms `bt_inv`.  That chain stays host-side torch and is
    SUPPLIED BY THE CALLER of fit_frame / fit_window as a callable (not part of the kernel path).  This
    class is the part of it that needs no skeleton model -- the global (palm) rigid motion and the object refinement
    (fitting_single.py:219-230) -- so that the drivers, tests and the bench have a complete chain to differentiate
    through:  joints' = R_palm (joints - root) + root + T_palm  <=>  bt_inv' = bt_inv0 G^-1,
    obj_r = rot6d(obj_rot) Ro_pred,  obj_t = To_pred + obj_trans.

    Parameters are `[data_num, ...]` blocks like fitting_video.py:159-176 (data_num = 1 for fitting_single)."""

    def __init__(self, bt_inv0, T_pose_21, joints0, Ro_pred, To_pred, obj_verts, device='cuda'):
        t = lambda x, *s: torch.as_tensor(x, dtype=torch.float32).to(device).reshape(*s)
        self.bt_inv0 = t(bt_inv0, -1, 21, 4, 4)
        n = self.bt_inv0.shape[0]
        self.joints0, self.Ro_pred, self.To_pred = t(joints0, n, 21, 3), t(Ro_pred, n, 3, 3), t(To_pred, n, 3)
        self.T_pose_21 = t(T_pose_21, -1, 21, 3)
        if self.T_pose_21.shape[0] != n:
            self.T_pose_21 = self.T_pose_21[:1].expand(n, 21, 3).contiguous()
        self.obj_verts = t(obj_verts, -1, 3)
        eye62 = torch.eye(3, device=device)[:, :2].expand(n, 3, 2).contiguous()
        self.obj_rot = torch.nn.Parameter(eye62.clone())
        self.obj_trans = torch.nn.Parameter(torch.zeros(n, 3, device=device))
        self.palm_rot = torch.nn.Parameter(eye62.clone())
        self.palm_trans = torch.nn.Parameter(torch.zeros(n, 3, device=device))

    def param_groups(self, video=False):
        """Learning rates of fitting_single.py:191-198 / fitting_video.py:177-184 for the parameters this chain has."""
        lr = (1e-4, 1e-4, 1e-4, 1e-4) if video else (5e-4, 5e-4, 5e-4, 3e-4)
        return [{'params': p, 'lr': l} for p, l in zip((self.obj_rot, self.obj_trans, self.palm_rot, self.palm_trans), lr)]

    def parameters(self):
        return [self.obj_rot, self.obj_trans, self.palm_rot, self.palm_trans]

    def __call__(self, index=None):
        idx = slice(None) if index is None else _index_tensor(self, index, self.bt_inv0.device)
        if self.bt_inv0.is_cuda:
            # one launch (hn_rigid_pose) instead of ~60 operators forward and ~100 backward; the vertex sets are not formed:
            # the losses that used them take the poses (step_loss, honerf_amd.pose.VertsLossFn)
            from .pose import RigidPoseFn
            F_ = self.joints0[idx].shape[0]
            params = torch.cat([self.obj_rot[idx].reshape(F_, 6), self.obj_trans[idx], self.palm_rot[idx].reshape(F_, 6), self.palm_trans[idx]], dim=1)
            out = RigidPoseFn.apply(params, self.bt_inv0[idx].contiguous(), self.joints0[idx].contiguous(), self.Ro_pred[idx].contiguous(),
                                    self.To_pred[idx].contiguous(), True)
            return {'bt_inv': out[:, :336].reshape(F_, 21, 4, 4), 'T_pose_21': self.T_pose_21[idx], 'joint_3d': out[:, 336:399].reshape(F_, 21, 3),
                    'joint3d_pred': self.joints0[idx], 'obj_r': out[:, 399:408].reshape(F_, 3, 3), 'obj_t': out[:, 408:411], 'joint_loss': out[:, 411],
                    'Ro_pred': self.Ro_pred[idx], 'To_pred': self.To_pred[idx], 'obj_verts': self.obj_verts}
        R_palm = rot6d_to_matrix(self.palm_rot[idx])                               # [F,3,3]
        root = self.joints0[idx][:, :1, :]
        joint_3d = (R_palm.unsqueeze(1) @ (self.joints0[idx] - root).unsqueeze(-1))[..., 0] + root + self.palm_trans[idx].unsqueeze(1)
        # G p = R_palm (p - root) + root + T  ->  G^-1 p = R_palm^T (p - root - T) + root
        Rt = R_palm.transpose(1, 2)
        Ginv = torch.zeros(R_palm.shape[0], 4, 4, device=R_palm.device)
        Ginv[:, :3, :3] = Rt
        Ginv[:, :3, 3] = root[:, 0] - (Rt @ (root[:, 0] + self.palm_trans[idx]).unsqueeze(-1))[..., 0]
        Ginv[:, 3, 3] = 1.0
        bt_inv = self.bt_inv0[idx] @ Ginv.unsqueeze(1)
        obj_r = rot6d_to_matrix(self.obj_rot[idx]) @ self.Ro_pred[idx]
        obj_t = self.To_pred[idx] + self.obj_trans[idx]
        pred_v = (obj_r.unsqueeze(1) @ self.obj_verts[None, :, :, None])[..., 0] + obj_t.unsqueeze(1)
        comp_v = (self.Ro_pred[idx].unsqueeze(1) @ self.obj_verts[None, :, :, None])[..., 0] + self.To_pred[idx].unsqueeze(1)
        return {'bt_inv': bt_inv, 'T_pose_21': self.T_pose_21[idx], 'joint_3d': joint_3d, 'joint3d_pred': self.joints0[idx], 'obj_r': obj_r, 'obj_t': obj_t,
                'pred_obj_v_w': pred_v, 'compare_obj_v_w': comp_v}


class HaloPoseChain:
    """The reference's own parameter set and pose chain (fitting_single.py:160-235): the six refine leaves
    obj_rot_refine, obj_trans_refine, palm_rot_refine, palm_trans_refine, joint_refine_angle, palm_refine_angle; the hand
    half -- convert_joints / transform_to_canonical / PoseConverter.get_refine_3d_joint / palm motion /
    PoseConverter.forward, ~4 000 torch operators in the reference -- is ONE launch here (honerf_amd.pose, hn_pose_chain) and
    one more for its backward pass; the object half (fitting_single.py:227-235) is the few torch operators it is there.
    Same interface as RigidPoseChain: a callable returning the renderer's pose inputs and the loss's joint / vertex sets.

    ori_3d_pose [n,21,3]: the predicted joints (MANO order); cur_bone_length [n,20]; T_pose_21 [n,21,3] or None: the rest
    position of every bone's joint -- None derives it from the chain's initial state (T_b = bt_inv0[b] joint_b, so that
    the bone-local coordinate of utils/fields.py:30-31 vanishes at joint b, the role the dataset's T_pose_21 plays)."""

    def __init__(self, ori_3d_pose, cur_bone_length, T_pose_21, Ro_pred, To_pred, obj_verts, device='cuda'):
        from .pose import PoseChainFn
        self._fn = PoseChainFn
        t = lambda x, *s: torch.as_tensor(x, dtype=torch.float32).to(device).reshape(*s)
        self.joints0 = t(ori_3d_pose, -1, 21, 3)
        n = self.joints0.shape[0]
        self.bone_len, self.Ro_pred, self.To_pred = t(cur_bone_length, n, 20), t(Ro_pred, n, 3, 3), t(To_pred, n, 3)
        self.obj_verts = t(obj_verts, -1, 3)
        eye62 = torch.eye(3, device=device)[:, :2].expand(n, 3, 2).contiguous()
        self.obj_rot = torch.nn.Parameter(eye62.clone())
        self.obj_trans = torch.nn.Parameter(torch.zeros(n, 3, device=device))
        self.palm_rot = torch.nn.Parameter(eye62.clone())
        self.palm_trans = torch.nn.Parameter(torch.zeros(n, 3, device=device))
        self.joint_refine_angle = torch.nn.Parameter(torch.zeros(n, 20, device=device))
        self.palm_refine_angle = torch.nn.Parameter(torch.zeros(n, 7, device=device))
        if T_pose_21 is None:
            with torch.no_grad():
                bt0, j0 = self._hand(slice(None))
                # element-wise (no BLAS call: a batched 3 x 3 product through the GEMM library may pick a reduced-precision
                # algorithm, and this derived constant enters every bone coordinate of the hand field)
                T_pose_21 = (bt0[..., :3, :3] * j0.unsqueeze(-2)).sum(-1) + bt0[..., :3, 3]
        self.T_pose_21 = t(T_pose_21, -1, 21, 3)
        if self.T_pose_21.shape[0] != n:
            self.T_pose_21 = self.T_pose_21[:1].expand(n, 21, 3).contiguous()

    LEAF_NAMES = ('obj_rot', 'obj_trans', 'palm_rot', 'palm_trans', 'joint_refine_angle', 'palm_refine_angle')

    @classmethod
    def stack(cls, chains):
        """ONE chain over the frames of several one-frame chains (fitting_single's independent frames fitted side by side,
        BatchedSingleFit): the constants and the current values of the six leaves row by row.  The frames stay independent problems
        -- every kernel of the chain works per frame -- and `unstack_into` writes the fitted rows back."""
        self = cls.__new__(cls)
        self._fn = chains[0]._fn
        cat = lambda name: torch.cat([getattr(c, name).detach() for c in chains], dim=0).contiguous()
        for name in ('joints0', 'bone_len', 'Ro_pred', 'To_pred', 'T_pose_21'):
            setattr(self, name, cat(name))
        same = all(c.obj_verts.data_ptr() == chains[0].obj_verts.data_ptr() or (c.obj_verts.shape == chains[0].obj_verts.shape and
                                                                              torch.equal(c.obj_verts, chains[0].obj_verts)) for c in chains[1:])
        self.obj_verts = chains[0].obj_verts
        self.obj_verts_per_frame = None if same else [c.obj_verts for c in chains]
        for name in cls.LEAF_NAMES:
            setattr(self, name, torch.nn.Parameter(cat(name)))
        return self

    def unstack_into(self, chains):
        with torch.no_grad():
            off = 0
            for c in chains:
                n = c.joints0.shape[0]
                for name in self.LEAF_NAMES:
                    getattr(c, name).copy_(getattr(self, name)[off:off + n])
                off += n

    def param_groups(self, video=False):
        """fitting_single.py:191-198; fitting_video.py:177-184: 1e-4 for all leaves but palm_refine_angle (5e-4)."""
        lr = (1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 5e-4) if video else (5e-4, 5e-4, 5e-4, 3e-4, 1e-3, 1e-3)
        return [{'params': p, 'lr': l} for p, l in zip(self.parameters(), lr)]

    def parameters(self):
        return [self.obj_rot, self.obj_trans, self.palm_rot, self.palm_trans, self.joint_refine_angle, self.palm_refine_angle]

    def _hand(self, idx):
        F = self.joints0[idx].shape[0]
        params = torch.cat([self.joint_refine_angle[idx], self.palm_refine_angle[idx], self.palm_rot[idx].reshape(F, 6), self.palm_trans[idx]], dim=1)
        return self._fn.apply(self.joints0[idx], self.bone_len[idx], params)

    def __call__(self, index=None):
        if index is None and self.joints0.is_cuda:
            # every frame of the chain (fitting_single: the one frame): one autograd node for the whole pose side
            from .pose import HaloChainFn
            bt_inv, joint_3d, obj_r, obj_t = HaloChainFn.apply(self.obj_rot, self.obj_trans, self.palm_rot, self.palm_trans,
                                                               self.joint_refine_angle, self.palm_refine_angle, self.joints0, self.bone_len,
                                                               self.Ro_pred, self.To_pred)
            return {'bt_inv': bt_inv, 'T_pose_21': self.T_pose_21, 'joint_3d': joint_3d, 'joint3d_pred': self.joints0, 'obj_r': obj_r, 'obj_t': obj_t,
                    'Ro_pred': self.Ro_pred, 'To_pred': self.To_pred, 'obj_verts': self.obj_verts}
        idx = slice(None) if index is None else _index_tensor(self, index, self.joints0.device)
        if self.joints0.is_cuda:
            # a window of frames (fitting_video): the same single node on the window's rows of the leaves; the window's constants
            # are gathered once per index set
            from .pose import HaloChainFn
            consts = (self.joints0, self.bone_len, self.Ro_pred, self.To_pred, self.T_pose_21)
            rows = index.tolist() if isinstance(index, torch.Tensor) else [int(i) for i in index]      # (a tensor index: one transfer)
            n_all = self.joints0.shape[0]
            if any(r < -n_all or r >= n_all for r in rows):
                raise IndexError('window %s outside the sequence of %d frames' % (rows, n_all))
            if any(r < 0 for r in rows):          # python's negative frame ids: as advanced indexing takes them
                rows = [r % n_all for r in rows]
                idx = _index_tensor(self, rows, self.joints0.device)
            # keyed on the constants' storage and version as well as on the rows: replacing or editing a constant drops its windows
            key = (tuple(rows), tuple((x.data_ptr(), x._version) for x in consts))
            cache = self.__dict__.setdefault('_window_consts', {})
            if key not in cache:
                if len(cache) > 4096:
                    cache.clear()
                cache[key] = tuple(x[idx].contiguous() for x in consts)
            j0, bl, Rp, Tp, T21 = cache[key]
            bt_inv, joint_3d, obj_r, obj_t = HaloChainFn.apply(self.obj_rot, self.obj_trans, self.palm_rot, self.palm_trans, self.joint_refine_angle,
                                                               self.palm_refine_angle, j0, bl, Rp, Tp, idx)
            return {'bt_inv': bt_inv, 'T_pose_21': T21, 'joint_3d': joint_3d, 'joint3d_pred': j0, 'obj_r': obj_r, 'obj_t': obj_t,
                    'Ro_pred': Rp, 'To_pred': Tp, 'obj_verts': self.obj_verts}
        bt_inv, joint_3d = self._hand(idx)
        obj_r = rot6d_to_matrix(self.obj_rot[idx]) @ self.Ro_pred[idx]
        obj_t = self.To_pred[idx] + self.obj_trans[idx]
        pred_v = (obj_r.unsqueeze(1) @ self.obj_verts[None, :, :, None])[..., 0] + obj_t.unsqueeze(1)
        comp_v = (self.Ro_pred[idx].unsqueeze(1) @ self.obj_verts[None, :, :, None])[..., 0] + self.To_pred[idx].unsqueeze(1)
        return {'bt_inv': bt_inv, 'T_pose_21': self.T_pose_21[idx], 'joint_3d': joint_3d, 'joint3d_pred': self.joints0[idx], 'obj_r': obj_r,
                'obj_t': obj_t, 'pred_obj_v_w': pred_v, 'compare_obj_v_w': comp_v}


def bone_lengths_of(joints_mano):
    """cur_bone_length for a set of MANO-order joints [n,21,3]: the 20 bone lengths in the biomech bone order the converter
    uses (kp3D_to_bones, halo_util/converter_fit_batch.py:537-562, on convert_joints(..., 'mano', 'biomech'))."""
    j = torch.as_tensor(joints_mano, dtype=torch.float32).reshape(-1, 21, 3)
    m2b = torch.tensor([0, 1, 5, 9, 13, 17, 2, 6, 10, 14, 18, 3, 7, 11, 15, 19, 4, 8, 12, 16, 20], device=j.device)
    kb = j[:, m2b]
    parent = torch.tensor([0] * 5 + list(range(1, 16)), device=j.device)
    return (kb[:, 1:] - kb[:, parent]).norm(dim=-1)


# ---- one optimisation step --------------------------------------------------------------------------------------------
def step_loss(render_out, true_rgb, true_mask, pose, fit_type='1', video=False, smooth_ends=(False, False), stable=None, pose_terms=None):
    """The full loss of one step: fitting_single.py:251-283 (video=False) or fitting_video.py:285-334 (video=True:
    + smoothness over the window's frames x50, anchored to the prediction at the sequence ends; + 100 x the stable
    term for fit type '1234').  `pose` is the pose chain's output dict."""
    if not video and render_out['color_fine'].is_cuda and 'obj_verts' in pose and pose['joint_3d'].shape[0] == 1:
        # fitting_single on the device: the whole loss is one autograd node over five launches (autograd.FitStepLossFn)
        from .autograd import FitStepLossFn
        interaction = fit_type in ('12', '123', '1234')
        weights = (1.0, 30.0, 20.0, 30.0, 20.0) if interaction else (1.0, 0.0, 0.0, 100.0, 5.0)
        loss, tv = FitStepLossFn.apply(render_out['color_fine'], render_out['weight_sum'], render_out['sdf_hand'] if interaction else None,
                                       render_out['sdf_obj'] if interaction else None, pose['joint_3d'], pose['obj_r'], pose['obj_t'], true_rgb,
                                       true_mask, pose['joint3d_pred'], pose['Ro_pred'], pose['To_pred'], pose['obj_verts'], weights)
        return {'loss': loss, 'color': tv[1], 'mask': tv[2], 'contact': tv[3], 'penetration': tv[4], 'joint': tv[5], 'obj_verts': tv[6]}
    if _fused_window_loss(render_out, pose, video):
        # fitting_video on the device: the whole loss of the window is one autograd node, one launch each way (autograd.FitWindowLossFn)
        from .autograd import FitWindowLossFn
        anchor = 1 if smooth_ends[0] else (2 if smooth_ends[1] else 0)
        term = stable if (stable is not None and not isinstance(stable, torch.Tensor)) else None       # autograd.StableTerm: the explicit form
        loss, tv = FitWindowLossFn.apply(render_out['color_fine'], render_out['weight_sum'], render_out['sdf_hand'], render_out['sdf_obj'], pose['joint_3d'],
                                         pose['obj_r'], pose['obj_t'], None if term is not None else stable, true_rgb, true_mask, pose['joint3d_pred'],
                                         pose['Ro_pred'], pose['To_pred'], pose['obj_verts'], anchor, pose['bt_inv'] if term is not None else None, term)
        terms = {'loss': loss, 'color': tv[1], 'mask': tv[2], 'contact': tv[3], 'penetration': tv[4], 'joint': tv[5], 'obj_verts': tv[6], 'smooth': tv[7]}
        if stable is not None:
            terms['stable'] = tv[8]
        return terms
    terms = render_loss_terms(render_out, true_rgb, true_mask, fit_type, video)
    pt = pose_terms if pose_terms is not None else pose_loss_terms(pose, fit_type, video, smooth_ends)
    terms['joint'], terms['obj_verts'] = pt['joint'], pt['obj_verts']
    terms['loss'] = terms['loss'] + pt['sum']
    if video:
        terms['smooth'] = pt['smooth']
        if stable is not None:
            terms['stable'] = 100.0 * stable
            terms['loss'] = terms['loss'] + terms['stable']
    return terms


DEFER_JACOBIAN = True      # fit_backward, window steps with the stable term: the chain's Jacobian launch behind the stable term's forward pass
FUSED_WINDOW_LOSS = True   # fitting_video on the device: the window's loss as one launch each way (False: the torch-operator form)


def _fused_window_loss(render_out, pose, video):
    return (FUSED_WINDOW_LOSS and video and render_out['color_fine'].is_cuda and 'obj_verts' in pose and 'sdf_hand' in render_out
            and 2 <= pose['joint_3d'].shape[0] <= 8)


def pose_loss_terms(pose, fit_type='1', video=False, smooth_ends=(False, False)):
    """The terms of a step's loss that depend on the pose chain alone (fitting_single.py:206-235, 283; fitting_video.py:310-334):
    the joint and object-vertex regularisers and, for a window, the smoothness term x50.  Returns them with 'sum' = their
    weighted sum as it enters the loss.  Independent of the render: the device loops evaluate them beside it (fit_backward)."""
    fused = 'obj_verts' in pose        # the chains' device form: vertex losses from the poses (VertsLossFn), no vertex sets
    if fused:
        from .pose import VertsLossFn
        v_pred = VertsLossFn.apply(pose['obj_r'], pose['obj_t'], pose['Ro_pred'], pose['To_pred'], pose['obj_verts'])   # [F]
    if not video:
        joint_loss = pose['joint_loss'][0] if 'joint_loss' in pose else pose_loss(pose['joint3d_pred'][0], pose['joint_3d'][0])
        verts_loss = v_pred[0] if fused else pose_loss(pose['compare_obj_v_w'][0], pose['pred_obj_v_w'][0])
        w = (100.0, 5.0) if fit_type == '1' else (30.0, 20.0)
    else:
        joint_loss = pose['joint_loss'].mean() if 'joint_loss' in pose else pose_loss(pose['joint_3d'], pose['joint3d_pred'], mean=True)
        verts_loss = v_pred.mean() if fused else pose_loss(pose['pred_obj_v_w'], pose['compare_obj_v_w'], mean=True)
        w = (30.0, 20.0)
    out = {'joint': joint_loss, 'obj_verts': verts_loss, 'sum': w[0] * joint_loss + w[1] * verts_loss}
    if video:
        j = pose['joint_3d']
        if fused:
            r, t_ = pose['obj_r'], pose['obj_t']
            smooth = pose_loss(j[1:], j[:-1], mean=True) + VertsLossFn.apply(r[1:], t_[1:], r[:-1], t_[:-1], pose['obj_verts']).mean()
            if smooth_ends[0]:
                smooth = smooth + pose_loss(j[:1], pose['joint3d_pred'][:1], mean=True) + v_pred[0]
            elif smooth_ends[1]:
                smooth = smooth + pose_loss(j[-1:], pose['joint3d_pred'][-1:], mean=True) + v_pred[-1]
        else:
            v = pose['pred_obj_v_w']
            smooth = pose_loss(j[1:], j[:-1], mean=True) + pose_loss(v[1:], v[:-1], mean=True)
            if smooth_ends[0]:
                smooth = smooth + pose_loss(j[:1], pose['joint3d_pred'][:1], mean=True) + pose_loss(v[:1], pose['compare_obj_v_w'][:1], mean=True)
            elif smooth_ends[1]:
                smooth = smooth + pose_loss(j[-1:], pose['joint3d_pred'][-1:], mean=True) + pose_loss(v[-1:], pose['compare_obj_v_w'][-1:], mean=True)
        out['smooth'] = 50.0 * smooth
        out['sum'] = out['sum'] + out['smooth']
    return out


def _rays(lib_mod, xy, cam, n_cams, rays_per_cam):
    lib = lib_mod.load()
    o = torch.empty(n_cams * rays_per_cam, 3, device=xy.device)
    d = torch.empty_like(o)
    lib_mod.check(lib.hn_ray_gen(lib_mod.ptr(xy), lib_mod.ptr(cam['R']), lib_mod.ptr(cam['T']), lib_mod.ptr(cam['focal']),
                                 lib_mod.ptr(cam['principal']), n_cams, rays_per_cam, lib_mod.ptr(o), lib_mod.ptr(d),
                                 lib_mod.stream_ptr()), 'hn_ray_gen')
    return o, d


PIPELINE_SINGLE = True    # fit_frame: the two-stream step (PipelinedSingleFit) where it applies; False: fit_backward + fit_apply through autograd
USE_SIDE_STREAM = True    # fit_backward (frame-batched renderer): the pose-only terms of a step on a second stream beside the render


def _side_stream(device):
    """ONE extra torch stream per device for branches of a step that are independent of the render -- the same stream the pose
    chain's Jacobian launches use (pose._aux_stream): with the caller's stream and the library's second stream that makes three.
    Every further stream of the process shares a hardware queue with one of those (the runtime multiplexes streams onto 4 queues),
    and work queued behind a 1 ms adjoint kernel of another stream is what that costs (measured: a stream per pipelined fit object
    took the 2.6 ms step to 4.4 ms by the third object)."""
    from .pose import _aux_stream
    return _aux_stream(device)


_UNIT = {}


def _unit_gradient(loss):
    """The 1 that `loss.backward()` starts from, kept per device: autograd otherwise allocates and fills one in every step (one more
    launch in a chain of dependent launches)."""
    if not loss.is_cuda or loss.dim() != 0 or loss.dtype != torch.float32:
        return None
    key = str(loss.device)
    if key not in _UNIT:
        _UNIT[key] = torch.ones((), device=loss.device, dtype=torch.float32)
    return _UNIT[key]


def fit_backward(renderer, view, pose_chain, near, far, fit_type='1', index=None, smooth_ends=(False, False),
                 obj_verts_for_stable=None, t_rand=None, rays_fn=None):
    """The forward + backward half of one optimiser step of the fitting loops: pose chain -> rays of the view's sampled
    pixels (`_xy_to_ray_bundle` -> hn_ray_gen) -> renderer.render -> losses -> backward into the pose parameters
    (fitting_single.py:201-290; batched renderer: fitting_video.py:212-341).  The parameters' `.grad` are cleared first
    (`optimizer.zero_grad()` of the reference) and hold this step's gradient afterwards; nothing is applied.

    view: dict(cam={'R','T','focal','principal'} device tensors [n_cams,..], xy [n_cams*P,2] NDC, true_rgb, true_mask)
    with n_cams = 1 for fitting_single and the window's 4 cameras for fitting_video.  `rays_fn(xy, cam, n_cams, P)`
    replaces hn_ray_gen (host-logic tests over a stub renderer; the product path never passes it)."""
    video = bool(getattr(renderer, 'batched', False))
    for p in pose_chain.parameters():
        p.grad = None
    # A window step with the stable term: the chain's Jacobian goes to the extra stream BEHIND the stable term's forward pass (its
    # value is what the loss waits for; the Jacobian is not read before the backward pass reaches the pose side).  It is also a
    # matter of correctness on this runtime (round 5, tools/seq_repro_diag.py, profiles/r05/seq_repro_*): the Jacobian kernel
    # keeps ~39 KB of private scratch per lane, for which the runtime re-sizes the queue's scratch at every launch, and a dispatch
    # queued on the SAME stream right behind it -- the stable term's first launches -- read its inputs' previous contents about once
    # in a hundred steps (hn_stable_pts against a torch restatement issued right behind it on the same stream: 3 mm apart, the
    # inputs equal).  Nothing is queued directly behind a Jacobian launch any more.
    defer = video and fit_type == '1234' and rays_fn is None and USE_SIDE_STREAM and DEFER_JACOBIAN
    if defer:
        from .pose import deferred_jacobians, flush_deferred_jacobians
        with deferred_jacobians():
            pose = pose_chain(index)
    else:
        pose = pose_chain(index)
    n_cams = view['cam']['R'].shape[0]
    P = view['xy'].shape[0] // n_cams
    if rays_fn is None:
        from . import lib as L
        rays_o, rays_d = _rays(L, view['xy'], view['cam'], n_cams, P)
    else:
        rays_o, rays_d = rays_fn(view['xy'], view['cam'], n_cams, P)
    T_pose = pose['T_pose_21']
    stable, pterms, side = None, None, None
    want_stable = video and fit_type == '1234'
    fused_loss = video and rays_o.is_cuda and FUSED_WINDOW_LOSS and 'obj_verts' in pose and 2 <= pose['joint_3d'].shape[0] <= 8
    pose_ready = None

    def pose_only_terms():
        """What depends on the pose only -- the stable term (the hand field's taped evaluation on the object's vertices: a few
        launches, but ~0.5 ms of one round of tiles forward and ~0.9 ms backward) and, in the torch-operator form of the loss, the
        pose regularisers / smoothness -- on a second stream beside the render; autograd runs their backward passes on that stream
        too, beside the render's.  They join the loss below."""
        nonlocal stable, pterms, side
        side = _side_stream(rays_o.device)
        side.wait_event(pose_ready)
        with torch.cuda.stream(side):
            if want_stable:
                # with the fused loss node: the explicit form (autograd.StableTerm) -- the loss node's backward, the first node of the
                # backward pass, queues the term's backward launches itself, ahead of the render's adjoint kernels
                stable = renderer.get_stable_loss_cross(obj_verts_for_stable, pose['bt_inv'], T_pose, pose['obj_r'], pose['obj_t'],
                                                        **({'as_term': True} if term_form else {}))
            if not fused_loss:
                pterms = pose_loss_terms(pose, fit_type, True, smooth_ends)
        for x in pose.values():
            if isinstance(x, torch.Tensor) and x.is_cuda:
                x.record_stream(side)
    use_side = video and rays_o.is_cuda and rays_fn is None and USE_SIDE_STREAM and (want_stable or not fused_loss)
    term_form = use_side and fused_loss and want_stable and obj_verts_for_stable is not None and \
        getattr(renderer, 'fused_stable_applies', lambda pts: False)(obj_verts_for_stable)
    if use_side:
        pose_ready = torch.cuda.Event()
        pose_ready.record()                      # the pose chain's outputs are complete here (the render's launches come after)
        pose_only_terms()
    if defer:
        flush_deferred_jacobians()
    if use_side:
        pass
    elif want_stable:
        stable = renderer.get_stable_loss_cross(obj_verts_for_stable, pose['bt_inv'], T_pose, pose['obj_r'], pose['obj_t'])
    if video:
        if rays_o.is_cuda and rays_fn is None:
            from .autograd import Mat3InverseFn
            Ro_arg = Mat3InverseFn.apply(pose['obj_r'])                                # fitting_video.py:284, one launch each way
        else:
            Ro_arg = torch.inverse(pose['obj_r'])                                      # fitting_video.py:284
        out = renderer.render(rays_o.reshape(n_cams, P, 3), rays_d.reshape(n_cams, P, 3), near, far, pose['bt_inv'], T_pose, None,
                              Ro_arg, pose['obj_t'], t_rand=t_rand)
    else:
        # frame 0 of a one-frame chain as a reshape: the backward of `x[0]` is a zero fill and a copy per tensor, that of a view is free
        one = pose['bt_inv'].shape[0] == 1
        first = lambda x: x.reshape(x.shape[1:]) if one else x[0]
        Ro_arg = first(pose['obj_r']).T                                                # fitting_single.py:250
        out = renderer.render(rays_o, rays_d, near, far, first(pose['bt_inv']), first(T_pose), None, Ro_arg, first(pose['obj_t']), t_rand=t_rand)
    if side is not None:
        main = torch.cuda.current_stream()
        main.wait_stream(side)
        shared = [stable] if isinstance(stable, torch.Tensor) else ([stable.value] if stable is not None else [])
        for x in shared + (list(pterms.values()) if pterms is not None else []):
            x.record_stream(main)
    terms = step_loss(out, view['true_rgb'], view['true_mask'], pose, fit_type, video, smooth_ends, stable, pterms)
    terms['loss'].backward(gradient=_unit_gradient(terms['loss']))
    return terms


def fit_apply(optimizer, pose_chain=None, dist=None, sync=False):
    """The second half of a step: with `sync`, one SUM all-reduce of the flattened pose-gradient block over all ranks
    (`allreduce_pose_gradients`; a rank that had no window this step contributes zeros), then the Adam step
    (fitting_single.py:291, fitting_video.py:342).  Returns the number of floats exchanged (0 without a collective).
    With sync every parameter receives a (possibly zero) gradient on every rank, so all replicas take the same step."""
    n = 0
    if sync:
        n = allreduce_pose_gradients(pose_chain.parameters(), dist)
    optimizer.step()
    return n


def fit_step(renderer, view, pose_chain, optimizer, near, far, fit_type='1', index=None, smooth_ends=(False, False),
             obj_verts_for_stable=None, t_rand=None, rays_fn=None, pipelined=False):
    """One optimiser step of the fitting loops on one rank: `fit_backward` then `fit_apply` without a collective
    (fitting_single.py:201-291; fitting_video.py:212-342).  pipelined: where it applies (fitting_single on the device, the
    reference's pose chain, PoseAdam) the step runs as PipelinedSingleFit -- same kernels, the hand's and the object's halves on two
    streams that stay apart across steps; call `finish_pipeline(optimizer)` (or synchronise the device) before reading the
    parameters.  `fit_frame` does both."""
    if pipelined and PipelinedSingleFit.applicable(renderer, pose_chain, optimizer, fit_type, index, rays_fn):
        return pipelined_fit(renderer, pose_chain, optimizer, near, far, fit_type).step(view, t_rand)
    terms = fit_backward(renderer, view, pose_chain, near, far, fit_type, index, smooth_ends, obj_verts_for_stable, t_rand, rays_fn)
    fit_apply(optimizer)
    return terms


def make_optimizer(pose_chain, video=False):
    """Adam over the chain's parameter groups with the learning rates of fitting_single.py:191-199 /
    fitting_video.py:177-185.  On the GPU the fused multi-tensor form: one kernel per step for all parameter blocks
    instead of ~10 element-wise launches per block (the step is a chain of dependent launches; every one counts)."""
    groups = pose_chain.param_groups(video=video)
    on_gpu = all(p.is_cuda for g in groups for p in ([g['params']] if isinstance(g['params'], torch.Tensor) else g['params']))
    if on_gpu and sum(len([g['params']] if isinstance(g['params'], torch.Tensor) else g['params']) for g in groups) <= 16:
        return PoseAdam(groups)
    return torch.optim.Adam(groups)


class PoseAdam:
    """torch.optim.Adam (its defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) over the few small
    pose-parameter blocks of a fitting loop as ONE launch per step (hn_adam_step).  The reference gives each of its six
    blocks its own learning rate (fitting_single.py:191-199), which as a torch optimiser is one fused launch per group plus
    the step counters -- a dozen dependent launches for 45 floats.  Same update formula, same skip of parameters whose
    `.grad` is None; `param_groups` / `zero_grad` / `step` as an Optimizer has them."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = []
        for g in groups:
            ps = [g['params']] if isinstance(g['params'], torch.Tensor) else list(g['params'])
            self.param_groups.append({'params': ps, 'lr': float(g['lr'])})
        self.betas, self.eps = betas, eps
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def add_param_group(self, group):
        ps = [group['params']] if isinstance(group['params'], torch.Tensor) else list(group['params'])
        self.param_groups.append({'params': ps, 'lr': float(group['lr'])})

    def state_dict(self):
        """torch.optim.Optimizer.state_dict's layout: parameters numbered in group order, per-parameter 'step' / 'exp_avg' /
        'exp_avg_sq' (what torch.optim.Adam checkpoints, so a state saved from either loads into the other)."""
        ids, groups, n = {}, [], 0
        # (the other keys a torch.optim.Adam group carries, at the defaults this optimiser implements)
        defaults = dict(torch.optim.Adam([torch.zeros(1)]).defaults)
        for g in self.param_groups:
            idx = []
            for p in g['params']:
                ids[id(p)] = n
                idx.append(n)
                n += 1
            groups.append(dict(defaults, lr=g['lr'], betas=tuple(self.betas), eps=self.eps, params=idx))
        state = {ids[k]: {'step': torch.tensor(float(st[2])), 'exp_avg': st[0].clone(), 'exp_avg_sq': st[1].clone()}
                 for k, st in self.state.items() if k in ids}
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != len(self.param_groups) or any(len(a['params']) != len(b['params']) for a, b in zip(groups, self.param_groups)):
            raise ValueError('PoseAdam.load_state_dict: the parameter groups do not match')
        params = [p for g in self.param_groups for p in g['params']]
        for g, src in zip(self.param_groups, groups):
            g['lr'] = float(src['lr'])
        if groups:
            self.betas, self.eps = tuple(groups[0].get('betas', self.betas)), float(groups[0].get('eps', self.eps))
        self.state = {}
        for k, st in sd['state'].items():
            p = params[int(k)]
            m = st['exp_avg'].to(device=p.device, dtype=torch.float32).reshape(p.shape).contiguous().clone()
            v = st['exp_avg_sq'].to(device=p.device, dtype=torch.float32).reshape(p.shape).contiguous().clone()
            self.state[id(p)] = [m, v, int(float(st['step']))]

    def ensure_state(self, p):
        """The moment buffers of `p`, created (zero) on torch's CURRENT stream if they do not exist yet.  A caller that steps on
        another stream (`step(stream=...)`) creates them beforehand and orders that stream behind the fills: created lazily inside
        such a step, the zero fills sat on the current stream behind a millisecond of queued kernels while the Adam launch on
        the other stream had already read -- and written -- the buffers (PipelinedSingleFit's first step, found by the
        bit-reproducibility test of round 4)."""
        st = self.state.get(id(p))
        if st is None:
            st = self.state[id(p)] = [torch.zeros_like(p, memory_format=torch.contiguous_format), torch.zeros_like(p, memory_format=torch.contiguous_format), 0]
        return st

    @torch.no_grad()
    def step(self, only=None, stream=None):
        """only: restrict the step to these parameters (the pipelined fitting step updates the hand's and the object's leaves on
        two streams); stream: the raw stream handle to launch on (default: torch's current stream)."""
        import ctypes
        from . import lib as L
        keep_ids = None if only is None else {id(p) for p in only}
        todo = [(p, g['lr']) for g in self.param_groups for p in g['params']
                if p.grad is not None and (keep_ids is None or id(p) in keep_ids)]
        if not todo:
            return
        if len(todo) > 16:      # (a launch takes 16 blocks: the networks of a training iteration, 43 tensors, go in three)
            for at in range(0, len(todo), 16):
                self._launch(todo[at:at + 16], stream)
            return
        self._launch(todo, stream)

    def _launch(self, todo, stream):
        import ctypes
        from . import lib as L
        n = len(todo)
        P, G, M, V = ((ctypes.c_void_p * n)() for _ in range(4))
        sizes, lrs, steps = (ctypes.c_int * n)(), (ctypes.c_float * n)(), (ctypes.c_int * n)()
        keep = []
        for i, (p, lr) in enumerate(todo):
            st = self.ensure_state(p)
            st[2] += 1                                   # torch counts the steps per parameter
            g = p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.contiguous().float()
            keep.append(g)
            assert p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda, 'PoseAdam: contiguous fp32 device parameters'
            P[i], G[i], M[i], V[i] = p.data_ptr(), g.data_ptr(), st[0].data_ptr(), st[1].data_ptr()
            sizes[i], lrs[i], steps[i] = p.numel(), lr, st[2]
        L.check(L.load().hn_adam_step(n, P, G, M, V, sizes, lrs, self.betas[0], self.betas[1], self.eps, steps,
                                      L.stream_ptr() if stream is None else stream), 'hn_adam_step')


class PipelinedSingleFit:
    """One optimisation step of fitting_single (fitting_single.py:201-291) as explicit launches on TWO streams that stay apart
    across steps -- the same kernels and the same arithmetic as `fit_backward` + `fit_apply`, without autograd in between -- for ONE
    frame or for SEVERAL independent frames side by side (a chain of F frames: `HaloPoseChain.stack`, `fit_frames_batched`).

    Why two streams: a step is `pose chain -> sampling -> both fields -> loss -> both adjoints -> Adam`, and the object's kernels are
    the long pole of both field phases (294 tiles on the CUs the hand leaves free: `k_field2_obj<4>` ends ~0.35 ms after
    `k_field2_hand<4>`).  But the hand's half of what follows -- its leaf gradients, Adam on its four leaves, the pose chain of the
    NEXT step (0.19 ms) and that step's hand sampling track (0.5 ms, the longer of the two tracks) -- needs nothing from the
    object's adjoint.  So the hand's half lives on the caller's stream and the object's half (its adjoint, its leaf gradients,
    Adam on the two object leaves, `hn_rigid_pose`, its local rays, its sampling track) on the library's second stream
    (`hn_side_stream`), and the two meet only where the data does: at the sort of the merged depths and at the compositing
    (`hn_render_dual` with HN_DUAL_OBJ_POSE_ON_SIDE, `hn_render_dual_bwd` with HN_DUAL_BWD_NO_JOIN).  Every buffer a step touches
    is owned here and lives as long as the object; the one buffer the next step's hand side would overwrite while the object's
    adjoint of this step still reads it -- the world rays -- is double-buffered.

    Why several frames: one frame's launches leave the chip half empty (a step's 294 object + ~160 live hand tiles are 1.8 rounds of
    256 one-workgroup CUs, its sampling rounds light 25 .. 100 CUs for 80 us each); frames of fitting_single are independent problems
    (own leaves, own Adam moments, fitting_single.py:134-199), so F of them go through the SAME launches -- rays [F x R], per-frame
    poses, per-frame losses with fitting_single's own normalisation, element-wise Adam over [F, .] leaf blocks.  Nothing a frame
    computes depends on its batch partners, to the BIT: per-ray / per-sample kernels do not look across rays, the hand's compacted
    list is frame-aligned (hn_api.hip, k_hand_compact_write), every sum over a frame's samples is formed from that frame's own tiles
    in an order relative to its first one (k_pose_part_reduce, k_obj_rays_bwd), and the loss kernels give every frame its own blocks and sums
    (tests/test_gpu_surface.py::test_sharded_frames_do_not_depend_on_the_sharding_nor_on_the_batch_partners).

    Call `finish()` before reading the parameters from another stream (or synchronise the device): the object's leaves are
    updated on the second stream."""

    JITTER_BLOCK = 64
    HAND_LEAVES = ('palm_rot', 'palm_trans', 'joint_refine_angle', 'palm_refine_angle')
    OBJ_LEAVES = ('obj_rot', 'obj_trans')
    MAX_FRAMES = 16

    @staticmethod
    def applicable(renderer, pose_chain, optimizer, fit_type, index, rays_fn, frames=1):
        from .renderer import NeuSRenderer_fitting
        return (isinstance(renderer, NeuSRenderer_fitting) and not renderer.batched and isinstance(pose_chain, HaloPoseChain)
                and isinstance(optimizer, PoseAdam) and index is None and rays_fn is None and fit_type in ('1', '12')
                and pose_chain.joints0.is_cuda and pose_chain.joints0.shape[0] == frames and 1 <= frames <= PipelinedSingleFit.MAX_FRAMES
                and (renderer.precision or 'f16x3') == 'f16x3' and renderer.n_importance > 0)

    def __init__(self, renderer, pose_chain, optimizer, near, far, fit_type):
        import ctypes
        from . import lib as L
        self.L, self.lib = L, L.load()
        self.ren, self.chain, self.opt = renderer, pose_chain, optimizer
        self.near, self.far, self.fit_type = float(near), float(far), fit_type
        dev = pose_chain.joints0.device
        self.dev = dev
        Fr = self.F = int(pose_chain.joints0.shape[0])
        sp = ctypes.c_void_p()
        L.check(self.lib.hn_side_stream(ctypes.byref(sp)), 'hn_side_stream')
        self.side_ptr = ctypes.c_void_p(sp.value)
        self.side = torch.cuda.ExternalStream(sp.value, device=dev)
        from .pose import bind_streams
        bind_streams(dev)
        self.aux = _side_stream(dev)                          # the pose chain's Jacobian (see step); one such stream per device
        self.ev_prm, self.ev_jac = torch.cuda.Event(), torch.cuda.Event()
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        z = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)
        self.prm_h, self.prm_o = e(Fr, 36), z(Fr, 18)
        self.bt, self.j3, self.jac_h = e(Fr, 21, 4, 4), e(Fr, 21, 3), e(Fr, 399, 36)
        self.out_o, self.jac_o = e(Fr, 412), e(Fr, 412, 18)
        if Fr == 1:
            self.obj_r, self.obj_t = self.out_o[:, 399:408], self.out_o[:, 408:411]          # views: contiguous 9 / 3 floats
        else:
            self.obj_r, self.obj_t = e(Fr, 9), e(Fr, 3)                                       # (copied out of the [F, 412] block on the second stream)
        self.g_loss = torch.ones(1, device=dev)
        self.g_bt, self.g_tp, self.g_Ro, self.g_To = e(Fr, 21, 4, 4), e(Fr, 21, 3), e(Fr, 3, 3), e(Fr, 3)
        # the loss node's per-frame blocks, one plane per quantity (frame f of each: base + f x width)
        self.sums, self.gj, self.gR, self.gt = e(Fr, 6), e(Fr, 63), e(Fr, 9), e(Fr, 3)
        self.gj_o, self.gR_o, self.gt_o = e(Fr, 63), e(Fr, 9), e(Fr, 3)
        self._rays = None
        self._n_rays = -1
        self._step = 0
        interaction = fit_type in ('12', '123', '1234')
        self.interaction = interaction
        self.w5 = (ctypes.c_float * 5)(*((1.0, 30.0, 20.0, 30.0, 20.0) if interaction else (1.0, 0.0, 0.0, 100.0, 5.0)))
        ch = pose_chain
        names = ('joint_refine_angle', 'palm_refine_angle', 'palm_rot', 'palm_trans', 'obj_rot', 'obj_trans')
        if Fr == 1:
            # the six leaves move into the two parameter blocks the chain kernels read (views of them from here on: same values, same
            # shapes, contiguous), so that a step does not gather them (a cat launch on each stream)
            self.g45 = z(1, 45)
            homes = {'joint_refine_angle': self.prm_h[:, 0:20], 'palm_refine_angle': self.prm_h[:, 20:27], 'palm_rot': self.prm_h[:, 27:33].view(1, 3, 2),
                     'palm_trans': self.prm_h[:, 33:36], 'obj_rot': self.prm_o[:, 0:6].view(1, 3, 2), 'obj_trans': self.prm_o[:, 6:9]}
            with torch.no_grad():
                for k, view in homes.items():
                    p = getattr(ch, k)
                    view.copy_(p.detach().reshape(view.shape))
                    p.data = view
            self._homes = tuple((getattr(ch, k), view.data_ptr()) for k, view in homes.items())
            g = self.g45
            self.g45_h = self.g45_o = g
            self.grads = {'obj_rot': g[:, 36:42].view(1, 3, 2), 'obj_trans': g[:, 42:45], 'palm_rot': g[:, 27:33].view(1, 3, 2), 'palm_trans': g[:, 33:36],
                          'joint_refine_angle': g[:, 0:20], 'palm_refine_angle': g[:, 20:27]}
        else:
            # F frames: the leaves are [F, .] blocks of their own (Adam runs over whole blocks); the chain's input rows are gathered
            # from them (one cat per stream and step) and the [F, 45] gradient rows scattered into six contiguous gradient blocks
            # (hn_leaf_rows_scatter, one launch per stream: the hand's four from its stream's block, the object's two from the other)
            self._homes = ()
            self.g45_h, self.g45_o = z(Fr, 45), z(Fr, 45)
            self.rows = torch.arange(Fr, device=dev, dtype=torch.long)
            self.gblk_h, self.gblk_o = z(Fr * 45), z(Fr * 45)

            def blocks(buf):      # hn_leaf_rows_scatter's layout: obj_rot, obj_trans, palm_rot, palm_trans, joint, palm_angle
                n = Fr
                return {'obj_rot': buf[:6 * n].view(n, 3, 2), 'obj_trans': buf[6 * n:9 * n].view(n, 3), 'palm_rot': buf[9 * n:15 * n].view(n, 3, 2),
                        'palm_trans': buf[15 * n:18 * n].view(n, 3), 'joint_refine_angle': buf[18 * n:38 * n].view(n, 20),
                        'palm_refine_angle': buf[38 * n:45 * n].view(n, 7)}
            bh, bo = blocks(self.gblk_h), blocks(self.gblk_o)
            self.grads = {k: (bo[k] if k in self.OBJ_LEAVES else bh[k]) for k in names}
            for k in names:
                p = getattr(ch, k)
                assert p.is_contiguous() and p.shape[0] == Fr, 'the leaves of a stacked chain are contiguous [F, .] blocks'
        for k in names:
            optimizer.ensure_state(getattr(ch, k))     # (before the second stream is ordered behind this one, below)
        self._jitter, self._jitter_at = None, 0
        self.jitter_states = None      # per-frame generator states (fit_frames_batched: every frame draws its own jitter stream)
        self.hand_params = [getattr(ch, k) for k in self.HAND_LEAVES]
        self.obj_params = [getattr(ch, k) for k in self.OBJ_LEAVES]
        self.verts = [ch.obj_verts] * Fr if getattr(ch, 'obj_verts_per_frame', None) is None else list(ch.obj_verts_per_frame)
        self._verts_p = self._verts_n = None     # (host arrays of the frames' vertex pointers / counts, made at the first step)
        # whatever the caller's stream has queued so far (the leaves' initial values, earlier steps through autograd) comes first
        L.check(self.lib.hn_stream_wait(self.side_ptr, L.stream_ptr()), 'hn_stream_wait')

    def _sized(self, R, S):
        if R == self._n_rays:
            return
        if self._rays is not None:
            # the previous step's object half (adjoint, Adam, VJP) may still be reading the buffers dropped below on the second
            # stream; they were allocated on the caller's stream, so the allocator could hand their blocks to the new tensors at once
            self.finish()
        dev = self.dev
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        N = self.F * R
        n = N * S
        self._n_rays = R
        self._rays = [(e(N, 3), e(N, 3)), (e(N, 3), e(N, 3))]
        self.color, self.wsum, self.gerr, self.z = e(N, 3), e(N, 1), e(2), e(N, S)
        self.sdf_h, self.sdf_o, self.grad_h, self.grad_o = e(n, 1), e(n, 1), e(n, 3), e(n, 3)
        self.gc, self.gw, self.gsh, self.gso = e(N, 3), e(N), e(n), e(n)

    def _draw_jitter(self, R):
        """JITTER_BLOCK steps of jitter numbers [block, F x R, 1] in one launch per frame (fresh uniform numbers every step, as
        torch.rand per render gives).  With per-frame generator states every frame continues ITS OWN stream -- the numbers a
        one-frame fit of that frame would draw after the same seed -- whatever its batch partners are."""
        if self.jitter_states is None:
            return torch.rand(self.JITTER_BLOCK, self.F * R, 1, device=self.dev)
        keep = torch.cuda.get_rng_state(self.dev)
        parts = []
        for f in range(self.F):
            torch.cuda.set_rng_state(self.jitter_states[f], self.dev)
            parts.append(torch.rand(self.JITTER_BLOCK, R, 1, device=self.dev))
            self.jitter_states[f] = torch.cuda.get_rng_state(self.dev)
        torch.cuda.set_rng_state(keep, self.dev)
        return parts[0] if self.F == 1 else torch.cat(parts, dim=1)

    @torch.no_grad()
    def step(self, view, t_rand=None):
        """-> the step's loss terms (device scalars, as fit_step returns them; [F] vectors for a chain of F > 1 frames).  The six
        leaves' .grad hold the step's gradient.  view: the step's pixels -- for F frames `cam` holds F cameras and xy / true_rgb /
        true_mask the F frames' R rows each, frame after frame (`stack_views`)."""
        L, lib, ren, ch = self.L, self.lib, self.ren, self.chain
        dev = self.dev
        Fr = self.F
        hand, obj = ren.fields()
        s, so = L.stream_ptr(), self.side_ptr
        n_cams = view['cam']['R'].shape[0]
        assert n_cams == Fr, 'fitting_single: one camera per frame and step'
        R = view['xy'].shape[0] // n_cams
        S = ren.n_samples + 2 * ren.n_importance
        N = Fr * R
        n = N * S
        self._sized(R, S)
        rays_o, rays_d = self._rays[self._step & 1]
        self._step += 1
        # ---- pose side: the hand's chain on the caller's stream, the object's on the second stream
        homed = Fr == 1 and all(p.data_ptr() == at for p, at in self._homes)      # (a caller that re-assigned a leaf's storage: gather as before)
        if not homed:
            torch.cat([ch.joint_refine_angle, ch.palm_refine_angle, ch.palm_rot.reshape(Fr, 6), ch.palm_trans], dim=1, out=self.prm_h)
        # (values and Jacobian as two launches: the render waits for the values, a third of the chain's time; the Jacobian is read
        # by this step's hn_pose_side_vjp, ~2 ms from here, and runs beside the sampling on a stream of its own)
        self.ev_prm.record()
        L.check(lib.hn_pose_chain(L.ptr(ch.joints0), L.ptr(ch.bone_len), None, L.ptr(self.prm_h), Fr, L.ptr(self.bt), L.ptr(self.j3), None, s), 'hn_pose_chain')
        self.aux.wait_event(self.ev_prm)
        L.check(lib.hn_pose_chain(L.ptr(ch.joints0), L.ptr(ch.bone_len), None, L.ptr(self.prm_h), Fr, None, None, L.ptr(self.jac_h), self.aux.cuda_stream),
                'hn_pose_chain')
        self.ev_jac.record(self.aux)
        if not homed:
            with torch.cuda.stream(self.side):
                torch.cat([ch.obj_rot.reshape(Fr, 6), ch.obj_trans], dim=1, out=self.prm_o[:, :9])
        L.check(lib.hn_rigid_pose(None, None, L.ptr(ch.Ro_pred), L.ptr(ch.To_pred), L.ptr(self.prm_o), Fr, 0, L.ptr(self.out_o), L.ptr(self.jac_o), so),
                'hn_rigid_pose')
        if Fr > 1:
            with torch.cuda.stream(self.side):
                self.obj_r.copy_(self.out_o[:, 399:408])
                self.obj_t.copy_(self.out_o[:, 408:411])
        cam = view['cam']
        L.check(lib.hn_ray_gen(L.ptr(view['xy']), L.ptr(cam['R']), L.ptr(cam['T']), L.ptr(cam['focal']), L.ptr(cam['principal']), Fr, R, L.ptr(rays_o),
                               L.ptr(rays_d), s), 'hn_ray_gen')
        # ---- the two-field render, taped; the object's pose comes from the second stream, as obj_r (its transpose is applied)
        if t_rand is None:
            if self._jitter is None or self._jitter_at >= self._jitter.shape[0] or self._jitter.shape[1] != N:
                self._jitter, self._jitter_at = self._draw_jitter(R), 0
            tr = self._jitter[self._jitter_at]
            self._jitter_at += 1
        else:
            tr = L.f32(t_rand, dev).reshape(N, 1)
        need = lib.hn_render_dual_workspace_bytes(hand.handle, obj.handle, N, ren.n_samples, ren.n_importance)
        from .renderer import _POISON
        if _POISON:
            self.finish()     # (the debug fill of workspace and tape runs on this stream: not under the previous step's object adjoint)
        ws = ren._ws.get(need, dev)
        tape_bytes = lib.hn_render_dual_tape_bytes(hand.handle, obj.handle, N, S)
        tape = ren._tape.get(tape_bytes, dev)
        flags = L.HN_DUAL_RO_TRANSPOSED | L.HN_DUAL_OBJ_POSE_ON_SIDE
        prev = getattr(ren, '_pending_aux', None)
        prev = prev() if prev is not None else None
        if prev is not None:
            prev.own()        # an autograd render of this renderer whose backward pass has not run yet: its arrays leave the tape first
        L.check(lib.hn_render_dual(hand.handle, obj.handle, L.ptr(rays_o), L.ptr(rays_d), L.ptr(tr), Fr, R, self.near, self.far, ren.n_samples,
                                   ren.n_importance, ren.up_sample_steps, L.ptr(self.bt), L.ptr(ch.T_pose_21), L.ptr(self.obj_r), L.ptr(self.obj_t), 0,
                                   L.ptr(self.color), L.ptr(self.wsum), L.ptr(self.sdf_h), L.ptr(self.sdf_o), L.ptr(self.grad_h), L.ptr(self.grad_o),
                                   L.ptr(self.gerr), L.ptr(self.z), L.ptr(ws), ws.numel(), L.ptr(tape), tape_bytes, flags, s), 'hn_render_dual')
        ren._last_z_raw = ren.last_z_vals = self.z
        ren._tape_serial = getattr(ren, '_tape_serial', 0) + 1     # (an autograd render's pending backward must not read this tape)
        # ---- the loss and its gradient w.r.t. the render outputs and the pose-side values: one launch each for all frames (every frame
        #      is its own problem with fitting_single's own normalisation: blockIdx.y = frame, and what a frame's blocks compute is what
        #      the launch of its one-frame fit computes)
        import ctypes
        terms = torch.empty(Fr, 8, device=dev, dtype=torch.float32)
        from .autograd import _loss_scratch
        nf = R * S
        scratch, sneed = _loss_scratch(lib, R, nf if self.interaction else 0, dev, frames=Fr)
        tm, tc = L.f32(view['true_mask'], dev).reshape(-1), L.f32(view['true_rgb'], dev).reshape(-1, 3)
        sh = L.ptr(self.sdf_h) if self.interaction else None
        so_ = L.ptr(self.sdf_o) if self.interaction else None
        if self._verts_p is None:
            self._verts_p = (ctypes.c_void_p * Fr)(*[v.data_ptr() for v in self.verts])
            self._verts_n = (ctypes.c_int * Fr)(*[int(v.shape[0]) for v in self.verts])
        L.check(lib.hn_fit_step_loss_frames(Fr, L.ptr(self.color), L.ptr(self.wsum), L.ptr(tc), L.ptr(tm), R, sh, so_, nf, L.ptr(self.j3), L.ptr(ch.joints0), 21,
                                            L.ptr(self.obj_r), L.ptr(self.obj_t), L.ptr(ch.Ro_pred), L.ptr(ch.To_pred), self._verts_p, self._verts_n, self.w5,
                                            L.ptr(scratch), sneed, L.ptr(self.sums), L.ptr(terms), L.ptr(self.gj), L.ptr(self.gR), L.ptr(self.gt), s),
                'hn_fit_step_loss_frames')
        L.check(lib.hn_fit_step_loss_bwd_frames(Fr, L.ptr(self.color), L.ptr(self.wsum), L.ptr(tc), L.ptr(tm), R, sh, so_, nf, L.ptr(self.sums), L.ptr(self.g_loss),
                                                self.w5, L.ptr(self.gj), L.ptr(self.gR), L.ptr(self.gt), 21, L.ptr(self.gc), L.ptr(self.gw),
                                                L.ptr(self.gsh) if self.interaction else None, L.ptr(self.gso) if self.interaction else None,
                                                L.ptr(self.gj_o), L.ptr(self.gR_o), L.ptr(self.gt_o), s), 'hn_fit_step_loss_bwd_frames')
        # ---- backward pass of the render: the hand's branch ends on s, the object's on the second stream (no join)
        aux_off = lib.hn_render_dual_tape_aux_offset(hand.handle, obj.handle, N, S)
        a = tape[aux_off:aux_off + 32 * n].view(torch.float32)
        rgb_h, rgb_o, al_h, al_o = a[:3 * n], a[3 * n:6 * n], a[6 * n:7 * n], a[7 * n:8 * n]
        bneed = lib.hn_render_dual_bwd_workspace_bytes(hand.handle, obj.handle, N, S)
        wsb = ren._ws_bwd.get(bneed, dev)
        sample_dist = float(torch.tensor((self.far - self.near) / ren.n_samples, dtype=torch.float32))
        flags = L.HN_DUAL_RO_TRANSPOSED | L.HN_DUAL_BWD_NO_JOIN
        gsh_p = L.ptr(self.gsh) if self.interaction else None
        gso_p = L.ptr(self.gso) if self.interaction else None
        L.check(lib.hn_render_dual_bwd(hand.handle, obj.handle, L.ptr(rays_o), L.ptr(rays_d), Fr, R, S, sample_dist, L.ptr(self.bt), L.ptr(ch.T_pose_21),
                                       L.ptr(self.obj_r), L.ptr(self.obj_t), L.ptr(self.z), L.ptr(self.sdf_h), L.ptr(self.grad_h), L.ptr(rgb_h), L.ptr(al_h),
                                       L.ptr(self.sdf_o), L.ptr(self.grad_o), L.ptr(rgb_o), L.ptr(al_o), L.ptr(self.gc), L.ptr(self.gw), gsh_p, gso_p, None, None,
                                       None, None, None, L.ptr(self.g_bt), L.ptr(self.g_tp), L.ptr(self.g_Ro), L.ptr(self.g_To), L.ptr(wsb), bneed, L.ptr(tape),
                                       flags, s), 'hn_render_dual_bwd')
        # ---- leaf gradients and Adam: the hand's four leaves on s, the object's two on the second stream
        torch.cuda.current_stream().wait_event(self.ev_jac)
        L.check(lib.hn_pose_side_vjp(L.ptr(self.jac_h), None, L.ptr(self.g_bt), L.ptr(self.gj_o), None, None, None, None, Fr, 1, L.ptr(self.g45_h), s),
                'hn_pose_side_vjp')
        L.check(lib.hn_pose_side_vjp(None, L.ptr(self.jac_o), None, None, L.ptr(self.g_Ro), L.ptr(self.g_To), L.ptr(self.gR_o), L.ptr(self.gt_o), Fr, 2,
                                     L.ptr(self.g45_o), so), 'hn_pose_side_vjp')
        if Fr > 1:
            L.check(lib.hn_leaf_rows_scatter(L.ptr(self.g45_h), L.ptr(self.rows), Fr, Fr, L.ptr(self.gblk_h), s), 'hn_leaf_rows_scatter')
            L.check(lib.hn_leaf_rows_scatter(L.ptr(self.g45_o), L.ptr(self.rows), Fr, Fr, L.ptr(self.gblk_o), so), 'hn_leaf_rows_scatter')
        for k, gview in self.grads.items():
            getattr(ch, k).grad = gview
        self.opt.step(only=self.hand_params, stream=s)
        self.opt.step(only=self.obj_params, stream=so)
        keys = ('loss', 'color', 'mask', 'contact', 'penetration', 'joint', 'obj_verts')
        if Fr == 1:
            return {k: terms[0, i] for i, k in enumerate(keys)}
        return {k: terms[:, i] for i, k in enumerate(keys)}

    def finish(self):
        """The caller's stream waits for the second stream's tail: from here on the parameters (all six leaves) and their .grad are
        visible to work queued on the caller's stream (and to a host read that synchronises with it)."""
        self.L.check(self.lib.hn_stream_wait(self.L.stream_ptr(), self.side_ptr), 'hn_stream_wait')


def ctypes_ptr(t, offset_floats=0):
    """Device address of element `offset_floats` of a contiguous fp32 tensor, as the C ABI takes it."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * int(offset_floats))


def stack_views(views):
    """The step's pixels of F frames as ONE view dict for PipelinedSingleFit.step: F cameras, the frames' xy / true_rgb / true_mask rows
    one frame after the other."""
    cam = {k: torch.cat([v['cam'][k] for v in views], dim=0).contiguous() for k in ('R', 'T', 'focal', 'principal')}
    cat = lambda k: torch.cat([v[k].reshape(-1, v[k].shape[-1]) for v in views], dim=0).contiguous()
    return {'cam': cam, 'xy': cat('xy'), 'true_rgb': cat('true_rgb'), 'true_mask': cat('true_mask')}


def pipelined_fit(renderer, pose_chain, optimizer, near, far, fit_type):
    """The PipelinedSingleFit of this (renderer, chain, optimiser), made on first use and kept on the optimiser."""
    key = (id(renderer), id(pose_chain), float(near), float(far), fit_type)
    cur = getattr(optimizer, '_pipeline', None)
    if cur is None or cur[0] != key:
        cur = (key, PipelinedSingleFit(renderer, pose_chain, optimizer, near, far, fit_type))
        optimizer._pipeline = cur
    return cur[1]


def finish_pipeline(optimizer):
    cur = getattr(optimizer, '_pipeline', None)
    if cur is not None:
        cur[1].finish()


def fit_frame(renderer, views, pose_chain, near, far, fit_type='1', n_iters=None, sample_view=None, rays_fn=None):
    """fitting_single.py:200-291 for one frame: `n_iters` (30 for fit type '1', 25 for '12'; 40 / 35 with 3 views,
    :124-132) passes over the views, one Adam step per view.  `sample_view(view_id, step) -> view dict` draws the
    step's pixels (the reference: get_rays_xy on the view's mask, 196 rays); default: the views as given."""
    if n_iters is None:
        n_iters = {('1', False): 30, ('1', True): 40, ('12', False): 25, ('12', True): 35}[(fit_type, len(views) == 3)]
    opt = make_optimizer(pose_chain, video=False)
    last, step = None, 0
    for _ in range(n_iters):
        for vid in range(len(views)):
            view = sample_view(vid, step) if sample_view is not None else views[vid]
            last = fit_step(renderer, view, pose_chain, opt, near, far, fit_type, rays_fn=rays_fn, pipelined=PIPELINE_SINGLE)
            step += 1
    finish_pipeline(opt)
    return last, step


FRAME_BATCH = int(os.environ.get('HONERF_FRAME_BATCH', '8'))   # frames a rank fits side by side when it owns several (fit_frames_sharded)


def frame_batch_capable(renderer, frame, fit_type, rays_fn=None):
    """Whether this (views, chain[, sample_view]) frame runs on the device path that fits several frames side by side."""
    try:
        from .renderer import NeuSRenderer_fitting
    except Exception:      # (no library: the host-logic tests over stub renderers)
        return False
    chain = frame[1]
    return (rays_fn is None and PIPELINE_SINGLE and isinstance(renderer, NeuSRenderer_fitting) and not renderer.batched and fit_type in ('1', '12')
            and renderer.n_importance > 0 and (renderer.precision or 'f16x3') == 'f16x3' and isinstance(chain, HaloPoseChain)
            and chain.joints0.is_cuda and chain.joints0.shape[0] == 1)


def frames_batchable(renderer, frames, fit_type, rays_fn=None):
    """Whether these (views, chain[, sample_view]) frames can go through ONE PipelinedSingleFit: the device path of fitting_single on
    the reference's pose chain, the same number of views and of pixels per view in every frame."""
    if len(frames) < 2 or len(frames) > PipelinedSingleFit.MAX_FRAMES or not all(frame_batch_capable(renderer, fr, fit_type, rays_fn) for fr in frames):
        return False
    v0 = frames[0][0]
    return all(len(fr[0]) == len(v0) and all(a['xy'].shape == b['xy'].shape and a['cam']['R'].shape[0] == 1 for a, b in zip(fr[0], v0)) for fr in frames)


def fit_frames_batched(renderer, frames, near, far, fit_type='12', n_iters=None, jitter_states=None):
    """fitting_single.py:200-291 for SEVERAL frames at once: `frames` = [(views, chain[, sample_view]), ...], every one its own
    optimisation problem (own six leaves, own Adam moments), all of them stepped through the SAME launches (PipelinedSingleFit over
    the stacked chain): per frame the same kernels on the same numbers as `fit_frame` of that frame alone, to the bit -- the batch is
    a way to fill the GPU when a rank owns more than one frame, not a change of the fit.  jitter_states: per frame the state of the
    CUDA generator its jitter is to be drawn from (what the global generator held when the frame's data was made: `fit_frame` draws
    from there), None: the global generator for all.  -> ([terms of the last step per frame], steps per frame)."""
    Fr = len(frames)
    chains = [fr[1] for fr in frames]
    n_views = len(frames[0][0])
    if n_iters is None:
        n_iters = {('1', False): 30, ('1', True): 40, ('12', False): 25, ('12', True): 35}[(fit_type, n_views == 3)]
    stacked = HaloPoseChain.stack(chains)
    opt = make_optimizer(stacked, video=False)
    assert PipelinedSingleFit.applicable(renderer, stacked, opt, fit_type, None, None, frames=Fr), 'frames_batchable() first'
    fit = PipelinedSingleFit(renderer, stacked, opt, near, far, fit_type)
    fit.jitter_states = list(jitter_states) if jitter_states is not None else None
    samplers = [fr[2] if len(fr) > 2 else None for fr in frames]
    static = None if any(sv is not None for sv in samplers) else [stack_views([fr[0][vid] for fr in frames]) for vid in range(n_views)]
    last, step = None, 0
    for _ in range(n_iters):
        for vid in range(n_views):
            if static is not None:
                view = static[vid]
            else:
                view = stack_views([(sv(vid, step) if sv is not None else fr[0][vid]) for sv, fr in zip(samplers, frames)])
            last = fit.step(view)
            step += 1
    fit.finish()
    stacked.unstack_into(chains)
    return [{k: v[f] for k, v in last.items()} for f in range(Fr)], step


def fit_window(renderer, views, pose_chain, optimizer, near, far, index, data_num, fit_type='1234', first_pass=False,
               obj_verts=None, sample_view=None, sub_iters=4, rays_fn=None, n_views=None):
    """fitting_video.py:211-342 for one window of 4 consecutive frames `index`: 4 sub-iterations x the views, an
    Adam step each on the shared [data_num, ..] parameters.  The smoothness term is anchored to the prediction when
    the window touches either end of the sequence (not on the very first step, :312)."""
    last, step = None, 0
    for sub in range(sub_iters):
        for vid in range(n_views if n_views is not None else len(views)):
            view = sample_view(vid, step) if sample_view is not None else views[vid]
            later = not (first_pass and sub == 0 and vid == 0)
            ends = (later and int(index[0]) == 0, later and int(index[-1]) == data_num - 1)
            last = fit_step(renderer, view, pose_chain, optimizer, near, far, fit_type, index=index, smooth_ends=ends,
                            obj_verts_for_stable=obj_verts, rays_fn=rays_fn)
            step += 1
    return last, step


def _dist_state(dist):
    """(dist module or None, rank, world) of the initialised process group; (None, 0, 1) without one."""
    if dist is None:
        import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def fit_sequence_video(renderer, window_views, pose_chain, near, far, data_num, fit_type='1234', outer_iters=5, sub_iters=4,
                       obj_verts=None, dist=None, optimizer=None, rays_fn=None, on_outer=None):
    """fitting_video.py:157, 186-342 for a whole sequence of `data_num` frames: `outer_iters` (5) passes over the
    data_num - 3 sliding windows, per window `sub_iters` (4) x the views, an Adam step each on the shared
    `[data_num, ..]` pose parameters of `pose_chain`.

    One rank: the reference's strictly sequential schedule.  `world` ranks: the windows of a pass are taken
    `world` at a time -- at round s rank r owns window s*world + r (`window_schedule`) -- and every one of the round's
    sub_iters x views steps is `fit_backward` on the rank's own window, ONE all-reduce (SUM) of the data_num x 45 pose
    gradient block, and the same Adam step on every replica (Jacobi over the `world` concurrent windows instead of
    Gauss-Seidel; a rank whose window list has run out contributes zeros).  Replicas stay bit-identical.

    window_views(index, view_id, step) -> the view dict of window `index` (list of 4 frame ids) for camera `view_id`
    (the reference draws 40 mask pixels per frame there, fitting_video.py:261-272); it must also report the number of
    views as `window_views.n_views`.  on_outer(iter_id) runs after every pass (the reference dumps all poses there,
    :350-425).  Returns dict(steps, windows, allreduce_calls, allreduce_floats, last=terms of the last step with a window)."""
    dist, rank, world = _dist_state(dist)
    if optimizer is None:
        optimizer = make_optimizer(pose_chain, video=True)
    sched = window_schedule(data_num, rank, world)
    n_views = window_views.n_views
    stats = {'steps': 0, 'windows': 0, 'allreduce_calls': 0, 'allreduce_floats': 0, 'last': None}
    for iter_id in range(outer_iters):
        for index in sched:
            step = 0
            for sub in range(sub_iters):
                for vid in range(n_views):
                    if index is not None:
                        # fitting_video.py:312, 316: anchored at the sequence ends, not on the very first step
                        later = iter_id + sub + vid > 0
                        ends = (later and index[0] == 0, later and index[-1] == data_num - 1)
                        stats['last'] = fit_backward(renderer, window_views(index, vid, step), pose_chain, near, far, fit_type,
                                                     index=index, smooth_ends=ends, obj_verts_for_stable=obj_verts, rays_fn=rays_fn)
                    else:
                        for p in pose_chain.parameters():
                            p.grad = None
                    n = fit_apply(optimizer, pose_chain, dist, sync=world > 1 or (dist is not None and FORCE_COLLECTIVE))
                    stats['allreduce_calls'] += int(n > 0)
                    stats['allreduce_floats'] += n
                    stats['steps'] += 1
                    step += 1
            stats['windows'] += int(index is not None)
        if on_outer is not None:
            on_outer(iter_id)
    return stats


def fit_frames_sharded(renderer, n_frames, make_frame, near, far, fit_type='12', n_iters=None, done=None, save=None, dist=None,
                       rays_fn=None, batch=None):
    """fitting_single.py:134-315 over a set of frames, sharded over the ranks: every frame is its own optimisation
    problem (own six parameters and Adam state, :177-199), so rank r fits frames r, r + world, .. with NO data-path
    collective; a frame whose result already exists is skipped (`done(frame_id)`, :156-158: a restarted or re-sharded
    run picks up what is missing, dealt out evenly over the ranks); `save(frame_id, pose_chain, terms)` is the pose dump of :293-315.  The only
    exchange is the SUM of the small loss vector at the end (`FrameShardedRunner.reduce`).

    batch (default FRAME_BATCH = 8): a rank that owns several frames fits up to `batch` of them SIDE BY SIDE through the same launches
    (`fit_frames_batched`) where the device path applies -- one frame's launches leave the GPU half empty -- with per-frame results
    that equal the one-by-one fits to the bit; 1: one by one.

    make_frame(frame_id) -> (views, pose_chain[, sample_view]).  Returns the reduced means + 'frames' + 'steps' of this rank."""
    dist, rank, world = _dist_state(dist)
    runner = FrameShardedRunner(n_frames, rank=rank, world=world, done=done)
    steps = [0]
    batch = FRAME_BATCH if batch is None else int(batch)

    def frame_fn(f):
        made = make_frame(f)
        views, chain = made[0], made[1]
        sample_view = made[2] if len(made) > 2 else None
        terms, n = fit_frame(renderer, views, chain, near, far, fit_type, n_iters, sample_view, rays_fn=rays_fn)
        steps[0] += n
        if save is not None:
            save(f, chain, terms)
        return terms

    def group_fn(fs):
        """-> the frames' terms, in order.  The frames' data are made one after the other; a frame's jitter is drawn from the generator
        state its make_frame left (where `fit_frame` of that frame alone would draw it from)."""
        first = make_frame(fs[0])
        if len(fs) == 1 or not frame_batch_capable(renderer, first, fit_type, rays_fn):
            # not the device path (or nothing to batch): strictly one after the other, every frame's data made right before its fit
            terms = []
            for k, f in enumerate(fs):
                m = first if k == 0 else make_frame(f)
                t, n = fit_frame(renderer, m[0], m[1], near, far, fit_type, n_iters, m[2] if len(m) > 2 else None, rays_fn=rays_fn)
                steps[0] += n
                if save is not None:
                    save(f, m[1], t)
                terms.append(t)
            return terms
        made, states = [], []
        for k, f in enumerate(fs):
            made.append(first if k == 0 else make_frame(f))
            states.append(torch.cuda.get_rng_state(made[-1][1].joints0.device))
        if frames_batchable(renderer, made, fit_type, rays_fn):
            terms, n = fit_frames_batched(renderer, made, near, far, fit_type, n_iters, jitter_states=states)
            steps[0] += n * len(fs)
        else:
            terms = []
            for f, m, st in zip(fs, made, states):
                if st is not None:
                    torch.cuda.set_rng_state(st, m[1].joints0.device)
                t, n = fit_frame(renderer, m[0], m[1], near, far, fit_type, n_iters, m[2] if len(m) > 2 else None, rays_fn=rays_fn)
                steps[0] += n
                terms.append(t)
        if save is not None:
            for f, m, t in zip(fs, made, terms):
                save(f, m[1], t)
        return terms

    out = runner.run(frame_fn) if batch <= 1 else runner.run_groups(group_fn, batch)
    out['steps'] = steps[0]
    out['frame_batch'] = batch
    out['rank_frames'] = list(runner.frames)
    out['allreduce_calls'] = getattr(runner, 'allreduce_calls', 0)
    return out


def synthetic_views(n_views, n_frames, rays_per_frame, seed, joints_center, device='cuda', H=230, W=266):
    """Synthetic stand-in for one fit_*_dataset item (the data set is an external download): `n_views` ring cameras
    looking at the hand, per view `rays_per_frame` pixels per frame drawn from a synthetic mask (NDC convention of
    utils/dataset.py:45-47), random target colours, mask = 1 inside.  n_frames > 1 lays the window's frames out as
    [F*P] rows with one camera per frame (all frames of a window share the view's camera, fitting_video.py:213-222)."""
    from . import synth
    cams = synth.ring_cameras(n_views, radius=0.9, target=tuple(joints_center), focal=2.0, seed=seed)
    g = torch.Generator('cpu').manual_seed(seed)
    views = []
    for v in range(n_views):
        cam = {k: torch.as_tensor(np.repeat(np.asarray(cams[k][v:v + 1], dtype=np.float32), n_frames, axis=0)).to(device).contiguous()
               for k in ('R', 'T', 'focal', 'principal')}
        xy = np.concatenate([synth.mask_pixels_ndc(H, W, rays_per_frame, seed * 1000 + 10 * v + f) * 0.25 for f in range(n_frames)])
        shape = (n_frames, rays_per_frame) if n_frames > 1 else (rays_per_frame,)
        views.append({'cam': cam, 'xy': torch.from_numpy(xy).to(device).contiguous(),
                      'true_rgb': torch.rand(*shape, 3, generator=g).to(device),
                      'true_mask': (torch.rand(*shape, 1, generator=g) > 0.2).float().to(device)})
    return views

"""Frame-sharded host logic of the fitting loops (fitting_single.py / fitting_video.py).

What is here is the part of the callers that decides WHICH frames / windows a rank works on, the
render-dependent loss terms they evaluate on the renderer's outputs, and the one reduction the
multi-GPU layout needs.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests); frames of `fitting_single` are independent optimisation problems
(own parameters and Adam state, fitting_single.py:143, 177-199), so they shard with NO data-path
collective -- the only exchange is the sum of a small loss/metric vector per logging interval.

`fitting_video` updates shared pose parameters window by window (fitting_video.py:146-149, 340-342);
`window_schedule` gives the synchronous window-parallel assignment of SURVEY 8(e) (rank r takes
window step*world + r); its gradient all-reduce belongs to the backward kernels.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

LOSS_KEYS = ('loss', 'color', 'mask', 'contact', 'penetration', 'frames')


def shard_frames(n_frames, rank, world):
    """Frames of this rank: strided assignment (frame f -> rank f % world), so that a growing
    sequence keeps every rank busy and a restart with another world size re-shards trivially."""
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


def render_loss_terms(render_out, true_rgb, true_mask, fit_type='1'):
    """The render-dependent loss terms of one optimisation step (fitting_single.py:251-283).

    color: L1(sum) of the masked colour error / B; mask: BCE(clip(weight_sum, 1e-3, 1-1e-3), mask),
    weighted 0.5; fit_type '12' adds contact (mean |s_h| + |s_o| where < 1e-2, x30) and penetration
    (mean over s_o < 0 and s_h < 0, x20).  The pose-regularisation terms (joint / object-vertex
    losses) depend on the pose chain only and stay with the caller."""
    color_fine = render_out['color_fine']
    weight_sum = render_out['weight_sum']
    color_error = (color_fine - true_rgb) * true_mask
    color_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / true_mask.shape[0]
    mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), true_mask)
    terms = {'color': color_loss, 'mask': mask_loss, 'loss': color_loss + 0.5 * mask_loss}
    zero = color_loss.new_zeros(())
    terms['contact'], terms['penetration'] = zero, zero
    if fit_type in ('12', '1234'):
        sdf_hand = render_out['sdf_hand'][:, 0]
        sdf_obj = render_out['sdf_obj'][:, 0]
        sdf_abs_sum = sdf_hand.abs() + sdf_obj.abs()
        contact_id = sdf_abs_sum < 1e-2
        contact = sdf_abs_sum[contact_id].sum() / (contact_id.float().sum() + 1e-9)
        inner = sdf_obj < 0
        hs, os_ = sdf_hand[inner], sdf_obj[inner]
        pen_id = hs < 0
        penet = (hs[pen_id].abs() + os_[pen_id].abs()).sum() / (pen_id.float().sum() + 1e-9)
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
        self.device = device or torch.device('cpu')
        self.frames = [f for f in shard_frames(n_frames, self.rank, self.world) if not (done and done(f))]
        self.totals = torch.zeros(len(LOSS_KEYS), dtype=torch.float64)

    def run(self, frame_fn):
        for f in self.frames:
            terms = frame_fn(f)
            for k, key in enumerate(LOSS_KEYS[:-1]):
                if key in terms:
                    self.totals[k] += float(terms[key])
            self.totals[-1] += 1
        return self.reduce()

    def reduce(self):
        t = self.totals.to(self.device)
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
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


def allreduce_pose_gradients(params, dist=None):
    """Window-parallel `fitting_video` step: every rank has back-propagated ITS window's loss into the
    shared `[data_num, ...]` pose parameters (non-zero on the window's 4 rows); one all-reduce (SUM) of the flattened
    gradient block -- data_num x 45 floats, ~18 KB at 100 frames: latency-bound, a single call -- makes the gradients
    identical on all ranks, after which every rank takes the same Adam step on its replica.  A rank whose window list
    has run out passes parameters without `.grad` (treated as zeros).  Returns the number of floats exchanged."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    params = list(params)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
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

"""Host logic of the frame-sharded fitting drivers (ho-nerf_amd/fitting.py) incl. the N > 1 path on
CPU: two processes, `gloo` backend, rendezvous on 127.0.0.1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from honerf_amd import fitting


def test_shards_partition_frames():
    for n in (0, 1, 7, 8, 33):
        for world in (1, 2, 4, 8):
            got = sorted(f for r in range(world) for f in fitting.shard_frames(n, r, world))
            assert got == list(range(n))
    with pytest.raises(ValueError):
        fitting.shard_frames(4, 2, 2)


def test_windows_match_reference_sampler():
    """RayImageSampler(N_images=4, N_iter=len-3) (utils/dataset.py:396-404, fitting_video.py:146-149)."""
    assert fitting.sliding_windows(6) == [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5]]
    assert fitting.sliding_windows(3) == []
    for world in (1, 2, 3):
        sched = [fitting.window_schedule(9, r, world) for r in range(world)]
        flat = [w for s in range(len(sched[0])) for r in range(world) for w in [sched[r][s]] if w is not None]
        assert flat == fitting.sliding_windows(9)     # step-major, rank-minor == the sequential order


def _fake_render(frame, rays=24, S=8):
    g = torch.Generator().manual_seed(100 + frame)
    return {
        'color_fine': torch.rand(rays, 3, generator=g), 'weight_sum': torch.rand(rays, 1, generator=g),
        'sdf_hand': (torch.rand(rays * S, 1, generator=g) - 0.5) * 0.05,
        'sdf_obj': (torch.rand(rays * S, 1, generator=g) - 0.5) * 0.05,
    }, torch.rand(rays, 3, generator=g), (torch.rand(rays, 1, generator=g) > 0.3).float()


def _frame_terms(frame):
    out, rgb, mask = _fake_render(frame)
    return fitting.render_loss_terms(out, rgb, mask, fit_type='12')


def test_loss_terms_formulas():
    """fitting_single.py:251-283 restated independently."""
    out, rgb, mask = _fake_render(3)
    t = fitting.render_loss_terms(out, rgb, mask, fit_type='12')
    color = ((out['color_fine'] - rgb) * mask).abs().sum() / mask.shape[0]
    w = out['weight_sum'].clip(1e-3, 1 - 1e-3)
    bce = -(mask * w.log() + (1 - mask) * (1 - w).log()).mean()
    sh, so = out['sdf_hand'][:, 0], out['sdf_obj'][:, 0]
    s = sh.abs() + so.abs()
    contact = s[s < 1e-2].sum() / ((s < 1e-2).float().sum() + 1e-9)
    pen = (sh < 0) & (so < 0)
    penet = s[pen].sum() / (pen.float().sum() + 1e-9)
    assert torch.allclose(t['color'], color) and torch.allclose(t['mask'], bce, atol=1e-6)
    assert torch.allclose(t['contact'], contact) and torch.allclose(t['penetration'], penet)
    assert torch.allclose(t['loss'], color + 0.5 * bce + 30 * contact + 20 * penet, atol=1e-6)
    t1 = fitting.render_loss_terms(out, rgb, mask, fit_type='1')
    assert float(t1['contact']) == 0.0 and torch.allclose(t1['loss'], color + 0.5 * bce, atol=1e-6)


def test_mask_pixels_convention():
    mask = np.zeros((8, 12), np.float32)
    mask[2:5, 3:9] = 1
    xy, idx = fitting.mask_pixels(mask, 50, np.random.default_rng(0))
    py, px = idx // 12, idx % 12
    assert mask[py, px].all()
    assert np.allclose(xy[:, 0], -(px - 6.0) / 4.0) and np.allclose(xy[:, 1], -(py - 4.0) / 4.0)


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        done = (lambda f: f == 5)            # a frame whose result already exists is skipped on restart
        runner = fitting.FrameShardedRunner(n_frames, done=done)
        q.put((rank, runner.frames, runner.run(_frame_terms)))
    finally:
        dist.destroy_process_group()


def test_two_rank_reduction_equals_single_process():
    n_frames = 11
    single = fitting.FrameShardedRunner(n_frames, rank=0, world=1, done=lambda f: f == 5).run(_frame_terms)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = sorted(f for _, fr, _ in res for f in fr)
    assert frames == [f for f in range(n_frames) if f != 5]
    for _, _, red in res:                      # both ranks hold the same reduced means
        assert red['frames'] == single['frames'] == n_frames - 1
        for k in fitting.LOSS_KEYS[:-1]:
            assert abs(red[k] - single[k]) < 1e-9, (k, red[k], single[k])


def test_to_image_matches_reference_formula():
    """exp_runner.py:370: (rgb * 255).clip(0, 255) on the [H, W, 3] reshape; uint8 as written by cv2.imwrite."""
    from honerf_amd.harness import to_image
    rgb = torch.tensor([[0.0, 0.5, 1.0], [1.2, -0.1, 0.999], [0.25, 0.75, 0.1], [0.0, 0.0, 0.0]])
    img = to_image(rgb, 2, 2)
    assert img.shape == (2, 2, 3) and img.dtype == np.uint8
    assert img[0, 0].tolist() == [0, 127, 255] and img[0, 1].tolist() == [255, 0, 254]


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_frames = 9
        torch.manual_seed(0)
        pose = [torch.zeros(n_frames, 6, requires_grad=True), torch.zeros(n_frames, 3, requires_grad=True),
                torch.zeros(n_frames, 20, requires_grad=True)]            # shared [data_num, .] parameters
        sched = fitting.window_schedule(n_frames, rank, world)
        target = torch.arange(n_frames, dtype=torch.float32)[:, None]
        log = []
        for win in sched:                                                 # one synchronous step per schedule entry
            for p in pose:
                p.grad = None
            if win is not None:
                loss = sum(((p[win] - target[win]) ** 2).sum() for p in pose)
                loss.backward()
            n = fitting.allreduce_pose_gradients(pose, dist)
            with torch.no_grad():
                for p in pose:
                    p -= 0.1 * p.grad
            log.append(n)
        q.put((rank, [p.detach().numpy().copy() for p in pose], log))     # (numpy: a tensor would travel as a file descriptor this process must outlive)
    finally:
        dist.destroy_process_group()


def test_window_parallel_gradient_allreduce_keeps_replicas_identical():
    """fitting_video's synchronous window-parallel step: after the all-reduce both ranks hold the same gradients, so
    their parameter replicas stay bit-identical, and equal a single process that sums the two windows' gradients."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, pose0, log0), (_, pose1, log1) = res
    pose0, pose1 = [torch.from_numpy(x) for x in pose0], [torch.from_numpy(x) for x in pose1]
    assert log0 == log1 and all(n == 9 * 29 for n in log0)
    for a, b in zip(pose0, pose1):
        assert torch.equal(a, b)
    # single-process reference of the same schedule
    n_frames = 9
    pose = [torch.zeros(n_frames, k) for k in (6, 3, 20)]
    target = torch.arange(n_frames, dtype=torch.float32)[:, None]
    s0, s1 = fitting.window_schedule(n_frames, 0, 2), fitting.window_schedule(n_frames, 1, 2)
    for w0, w1 in zip(s0, s1):
        grads = [torch.zeros_like(p) for p in pose]
        for win in (w0, w1):
            if win is not None:
                for g, p in zip(grads, pose):
                    g[win] += 2 * (p[win] - target[win])
        for g, p in zip(grads, pose):
            p -= 0.1 * g
    for a, b in zip(pose0, pose):
        assert torch.allclose(a, b, atol=1e-6)


def test_bench_gpus_flag_spawns_ranks_or_refuses_a_mismatch():
    """bench.py --gpus N: without a torch.distributed environment it builds the driver's launch line for N ranks (run
    as a child process before any GPU call); inside one it refuses a world size that differs from --gpus instead of
    quietly reporting n_gpus = 1."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.spawn_command(4, ['--gpus', '4', '--steps', '2'])
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert '127.0.0.1' in cmd and cmd[-4:] == ['--gpus', '4', '--steps', '2'] and cmd[-5].endswith('bench.py')
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE=2' in r.stderr


def _train_grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from honerf_amd import training
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                                              # identical replicas of the parameters
        params = [torch.nn.Parameter(torch.randn(5, 7)), torch.nn.Parameter(torch.randn(5, 1)), torch.nn.Parameter(torch.randn(()))]
        opt = torch.optim.Adam(params, lr=1e-2)
        torch.manual_seed(100 + rank)                                     # a different "ray batch" on every rank
        for _ in range(3):
            x = torch.randn(11, 7)
            loss = ((x @ params[0].T * params[1].T).sum(dim=1) * params[2]).pow(2).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            training.allreduce_gradients(params, dist)
            opt.step()
        q.put((rank, [p.detach().numpy().copy() for p in params], [p.grad.numpy().copy() for p in params]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_training_gradients_are_averaged_and_replicas_identical():
    """honerf_amd.training.allreduce_gradients: one all-reduce of the flattened gradient block; both ranks end up with
    the same (averaged) gradients and therefore bit-identical parameters after Adam."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0), (_, p1, g1) = res
    for a, b in zip(p0 + g0, p1 + g1):
        assert np.array_equal(a, b)


# ---- the real fitting loops under two ranks (stub renderer, CPU pose chain) -------------------------------------------
class _StubRenderer:
    """A few differentiable torch operators with NeuSRenderer_fitting.render's signature and return keys
    (utils/renderer_batch.py:184-281): enough for the loop, loss and all-reduce code to run on the CPU."""
    batched = True
    S = 6

    def render(self, rays_o, rays_d, near, far, bt_inv, T_pose, verts, Ro, To, t_rand=None):
        z = torch.linspace(near, far, self.S)
        pts = rays_o[:, :, None, :] + rays_d[:, :, None, :] * z[None, None, :, None]              # [F,P,S,3]
        q = (bt_inv[:, None, None, 9, :3, :3] @ pts[..., None])[..., 0] + bt_inv[:, None, None, 9, :3, 3] - T_pose[:, None, None, 9]
        sdf_h = q.norm(dim=-1) - 0.05
        po = (Ro[:, None, None] @ (pts - To[:, None, None, :])[..., None])[..., 0]
        sdf_o = po.norm(dim=-1) - 0.03
        w_h, w_o = torch.sigmoid(-20 * sdf_h), torch.sigmoid(-20 * sdf_o)
        color = (w_h[..., None] * torch.sigmoid(q) + w_o[..., None] * torch.sigmoid(po)).mean(2)
        wsum = (0.5 * (w_h + w_o)).mean(2, keepdim=True)
        return {'color_fine': color, 'weight_sum': wsum, 'sdf_hand': sdf_h.reshape(-1, 1), 'sdf_obj': sdf_o.reshape(-1, 1)}

    def get_stable_loss_cross(self, obj_verts, bt_inv, T_pose, obj_r, obj_t):
        v = (obj_r[:, None] @ obj_verts[..., None])[..., 0] + obj_t[:, None]                        # [F,V,3]
        q = (bt_inv[:, None, 9, :3, :3] @ v[..., None])[..., 0] + bt_inv[:, None, 9, :3, 3]
        d = q.norm(dim=-1)
        return (d[1:] - d[:-1]).abs().mean()


def _stub_rays(xy, cam, n_cams, P):
    from oracle import render as orr
    o, d = zip(*[orr.rays_from_xy(xy[c * P:(c + 1) * P], cam['R'][c], cam['T'][c], cam['focal'][c], cam['principal'][c]) for c in range(n_cams)])
    return torch.cat(o), torch.cat(d)


_SEQ = dict(data_num=8, n_views=2, rays=5, outer=2, sub=2)


def _sequence_problem():
    from honerf_amd import synth
    n = _SEQ['data_num']
    rng = np.random.RandomState(5)
    bt, tp, j = synth.synth_hand_pose(3)
    R, tt = synth.synth_obj_pose(4, center=tuple(j[9] + np.array([0.02, 0.0, 0.01])))
    rep = lambda a: np.repeat(a[None], n, 0) + 0.002 * rng.standard_normal((n,) + a.shape).astype(np.float32)
    u = rng.standard_normal((40, 3))
    verts = (u / np.linalg.norm(u, axis=1, keepdims=True) * 0.025).astype(np.float32)
    chain = fitting.RigidPoseChain(np.repeat(bt[None], n, 0), np.repeat(tp[None], n, 0), rep(j), np.repeat(R[None], n, 0), rep(tt), verts, device='cpu')
    per_window = {}

    def window_views(index, vid, step):
        key = tuple(index)
        if key not in per_window:
            per_window[key] = fitting.synthetic_views(_SEQ['n_views'], 4, _SEQ['rays'], 100 + index[0], j[9], device='cpu')
        return per_window[key][vid]
    window_views.n_views = _SEQ['n_views']
    ov = torch.from_numpy(verts)[None].expand(4, -1, -1).contiguous()
    return chain, window_views, ov


def _run_sequence(dist=None):
    chain, window_views, ov = _sequence_problem()
    stats = fitting.fit_sequence_video(_StubRenderer(), window_views, chain, 0.4, 1.5, _SEQ['data_num'], '1234', outer_iters=_SEQ['outer'],
                                       sub_iters=_SEQ['sub'], obj_verts=ov, dist=dist, rays_fn=_stub_rays)
    return [p.detach().clone() for p in chain.parameters()], stats


def _sequence_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        params, stats = _run_sequence(dist)
        q.put((rank, [p.numpy() for p in params], {k: v for k, v in stats.items() if k != 'last'}))
    finally:
        dist.destroy_process_group()


def _jacobi_single_process(world):
    """The same synchronous schedule in one process: per step the gradients of the `world` concurrent windows are
    summed (rank order) before one Adam step."""
    chain, window_views, ov = _sequence_problem()
    opt = fitting.make_optimizer(chain, video=True)
    n = _SEQ['data_num']
    scheds = [fitting.window_schedule(n, r, world) for r in range(world)]
    ren = _StubRenderer()
    for iter_id in range(_SEQ['outer']):
        for s in range(len(scheds[0])):
            step = 0
            for sub in range(_SEQ['sub']):
                for vid in range(_SEQ['n_views']):
                    total = [torch.zeros_like(p) for p in chain.parameters()]
                    for r in range(world):
                        index = scheds[r][s]
                        if index is None:
                            continue
                        later = iter_id + sub + vid > 0
                        fitting.fit_backward(ren, window_views(index, vid, step), chain, 0.4, 1.5, '1234', index=index,
                                             smooth_ends=(later and index[0] == 0, later and index[-1] == n - 1),
                                             obj_verts_for_stable=ov, rays_fn=_stub_rays)
                        for t, p in zip(total, chain.parameters()):
                            if p.grad is not None:
                                t += p.grad
                    for t, p in zip(total, chain.parameters()):
                        p.grad = t
                    opt.step()
                    step += 1
    return [p.detach().clone() for p in chain.parameters()]


def test_fit_sequence_video_two_ranks_match_the_jacobi_schedule():
    """fit_sequence_video (the loop bench.py --gpus N runs) under 2 gloo ranks: one all-reduce per step, replicas
    bit-identical, and equal to a single process that sums the two concurrent windows' gradients per step; with one
    rank the loop is the reference's sequential schedule (fitting_video.py:186-342) = fit_window per window."""
    torch.set_num_threads(1)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sequence_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, st0), (_, p1, st1) = res
    n_win = _SEQ['data_num'] - 3                       # 5 windows -> 3 rounds of 2 ranks, the last half empty
    rounds = (n_win + 1) // 2
    steps = _SEQ['outer'] * rounds * _SEQ['sub'] * _SEQ['n_views']
    assert st0['steps'] == st1['steps'] == steps
    assert st0['allreduce_calls'] == st1['allreduce_calls'] == steps
    assert st0['allreduce_floats'] == steps * _SEQ['data_num'] * 18      # the rigid chain's 18 floats per frame
    assert st0['windows'] + st1['windows'] == _SEQ['outer'] * n_win
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)                    # replicas bit-identical
    ref = _jacobi_single_process(2)
    moved = 0.0
    for a, b in zip(p0, ref):
        assert np.allclose(a, b.numpy(), rtol=0, atol=1e-7), np.abs(a - b.numpy()).max()
        moved = max(moved, float(np.abs(a - np.round(a)).max()))
    assert moved > 1e-4                                # the parameters did move away from their identity / zero start
    # world = 1: the loop is the sequential schedule, i.e. fit_window per window with one optimiser
    seq, st = _run_sequence(None)
    assert st['allreduce_calls'] == 0 and st['windows'] == _SEQ['outer'] * n_win
    chain, window_views, ov = _sequence_problem()
    opt = fitting.make_optimizer(chain, video=True)
    for iter_id in range(_SEQ['outer']):
        for index in fitting.sliding_windows(_SEQ['data_num']):
            fitting.fit_window(_StubRenderer(), None, chain, opt, 0.4, 1.5, index, _SEQ['data_num'], '1234', first_pass=iter_id == 0, obj_verts=ov,
                               sample_view=lambda vid, step, index=index: window_views(index, vid, step), sub_iters=_SEQ['sub'],
                               rays_fn=_stub_rays, n_views=_SEQ['n_views'])
    for a, b in zip(seq, chain.parameters()):
        assert torch.equal(a, b.detach())


def test_grad_block_is_the_cat_of_the_gradients_or_nothing():
    """allreduce_pose_gradients reduces the leaves' gradients IN PLACE when they are views of one block laid out in parameter
    order (what HaloChainFn.backward hands out for a window) -- and only then: any other layout goes through cat / copy."""
    n = 5
    sizes = (6, 3, 6, 3, 20, 7)
    ps = [torch.zeros(n, k, requires_grad=True) for k in sizes]
    block = torch.arange(n * 45, dtype=torch.float32)
    off = 0
    for p, k in zip(ps, sizes):
        p.grad = block[off:off + n * k].view(n, k)
        off += n * k
    flat = fitting._grad_block(ps)
    assert flat is not None and flat.data_ptr() == block.data_ptr() and torch.equal(flat, torch.cat([p.grad.reshape(-1) for p in ps]))
    flat += 1.0                                   # in place: the leaves' gradients see it
    assert float(ps[3].grad[0, 0]) == float(block[15 * n])
    # a block whose storage order is not the parameter order is not "the cat": refused
    ps[0].grad, ps[2].grad = ps[2].grad, ps[0].grad
    assert fitting._grad_block(ps) is None
    ps[0].grad, ps[2].grad = ps[2].grad, ps[0].grad
    # a gap, a missing gradient, separate allocations: refused
    g = ps[5].grad
    ps[5].grad = g.clone()
    assert fitting._grad_block(ps) is None
    ps[5].grad = None
    assert fitting._grad_block(ps) is None
    ps[5].grad = g
    assert fitting._grad_block(ps[1:]) is not None      # (a sub-range that starts inside the block is still contiguous)


def test_forced_collective_on_a_one_rank_group_is_the_identity(monkeypatch):
    """HONERF_FORCE_COLLECTIVE: the sharded loops' collectives are issued on a group of ONE rank too (the single-GPU RCCL check of
    tests/test_gpu_surface.py / bench.py --rccl-one-rank); here over gloo: the count says they ran, the values say identity."""
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    try:
        ps = [torch.zeros(4, k, requires_grad=True) for k in (6, 3)]
        block = torch.randn(4 * 9)
        ps[0].grad, ps[1].grad = block[:24].view(4, 6), block[24:].view(4, 3)
        before = block.clone()
        assert fitting.allreduce_pose_gradients(ps, dist) == 0                     # one rank, not forced: no collective
        assert fitting.allreduce_pose_gradients(ps, dist, force=True) == 36        # forced: the block, in place
        assert torch.equal(block, before)
        ps[1].grad = ps[1].grad.clone()                                            # separate allocations: the cat / copy path
        assert fitting.allreduce_pose_gradients(ps, dist, force=True) == 36
        assert torch.equal(torch.cat([p.grad.reshape(-1) for p in ps]), before)
        plain = fitting.FrameShardedRunner(5, rank=0, world=1).run(_frame_terms)
        monkeypatch.setattr(fitting, 'FORCE_COLLECTIVE', True)
        runner = fitting.FrameShardedRunner(5)
        forced = runner.run(_frame_terms)
        assert runner.allreduce_calls == 1 and forced == plain
        out = _run_sequence(dist)
        assert out[1]['allreduce_calls'] == out[1]['steps'] > 0
        monkeypatch.setattr(fitting, 'FORCE_COLLECTIVE', False)
        ref = _run_sequence(None)
        assert ref[1]['allreduce_calls'] == 0
        for a, b in zip(out[0], ref[0]):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def _frames_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, _run_frames(dist, tmp)))
    finally:
        dist.destroy_process_group()


class _StubSingle(_StubRenderer):
    batched = False

    def render(self, rays_o, rays_d, near, far, bt_inv, T_pose, verts, Ro, To, t_rand=None):
        out = super().render(rays_o[None], rays_d[None], near, far, bt_inv[None], T_pose[None], None, Ro[None], To[None])
        return {k: (v[0] if k in ('color_fine', 'weight_sum') else v) for k, v in out.items()}


def _run_frames(dist, tmp):
    from honerf_amd import synth

    def make_frame(f):
        bt, tp, j = synth.synth_hand_pose(10 + f)
        R, tt = synth.synth_obj_pose(20 + f, center=tuple(j[9] + np.array([0.02, 0.0, 0.01])))
        verts = (np.random.RandomState(f).standard_normal((30, 3)) * 0.02).astype(np.float32)
        chain = fitting.RigidPoseChain(bt[None], tp[None], j[None], R[None], tt[None], verts, device='cpu')
        return fitting.synthetic_views(2, 1, 7, 50 + f, j[9], device='cpu'), chain

    def done(f):
        return os.path.exists(os.path.join(tmp, 'pose_%d.npy' % f))

    def save(f, chain, terms):
        np.save(os.path.join(tmp, 'pose_%d.npy' % f), np.concatenate([p.detach().reshape(-1).numpy() for p in chain.parameters()]))

    return fitting.fit_frames_sharded(_StubSingle(), 5, make_frame, 0.4, 1.5, '12', n_iters=2, done=done, save=save, dist=dist, rays_fn=_stub_rays)


def test_fit_frames_sharded_two_ranks_and_restart(tmp_path):
    """fit_frames_sharded (fitting_single.py:134-315 over frames): two ranks fit disjoint frames, the reduced loss
    means equal the 1-rank run's, the per-frame results are identical files, and a second run skips what exists."""
    torch.set_num_threads(1)
    one = str(tmp_path / 'one')
    two = str(tmp_path / 'two')
    os.makedirs(one)
    os.makedirs(two)
    np.save(os.path.join(one, 'pose_3.npy'), np.zeros(1))           # frame 3 "already fitted": skipped (fitting_single.py:156-158)
    np.save(os.path.join(two, 'pose_3.npy'), np.zeros(1))
    single = _run_frames(None, one)
    assert single['frames'] == 4 and single['steps'] == 4 * 2 * 2 and single['rank_frames'] == [0, 1, 2, 4]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_frames_worker, args=(r, 2, port, q, two)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the four frames still to do (3 exists) are dealt out AFTER the done filter: two per rank, not [0, 2, 4] / [1]
    assert res[0][1]['rank_frames'] == [0, 2] and res[1][1]['rank_frames'] == [1, 4]
    for _, red in res:
        assert red['frames'] == 4
        for k in fitting.LOSS_KEYS[:-1]:
            assert abs(red[k] - single[k]) < 1e-6, (k, red[k], single[k])
    for f in (0, 1, 2, 4):
        assert np.array_equal(np.load(os.path.join(one, 'pose_%d.npy' % f)), np.load(os.path.join(two, 'pose_%d.npy' % f)))
    again = _run_frames(None, one)
    assert again['frames'] == 0 and again['steps'] == 0

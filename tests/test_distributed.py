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
        q.put((rank, [p.detach().clone() for p in pose], log))
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

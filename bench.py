#!/usr/bin/env python3
"""bench.py -- ray-samples/sec of the HO-NeRF rendering hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): single-GPU offline render with the
hand nets at conf size, 512 x 512 rays x 64 samples per ray (n_samples=64,
n_importance=0), synthetic pose / camera / random-init weights, dense (no far-field
culling).  Arithmetic: "f16x3" -- fp16 hi/lo split operands, 3 x v_mfma_f32_32x32x16_f16 per
product, fp32 accumulate: fp32-equivalent results (parity ~3e-6 against the reference);
--precision fp32 selects the exact-fp32 MFMA kernels instead.  One step = ray generation + NeuSRenderer.render of one full frame with
everything already resident in HBM.  With --gpus N (launched by torch.distributed.run,
one rank per GPU) every rank renders its own frame (frame-sharded, no data-path
collective): weak scaling, value = all ranks' ray-samples / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel: the fused
hand field kernel, MFMA-bound), `cpu_baseline` (the CPU oracle -- a port of the
reference's PyTorch path -- timed on this box's host cores on a bounded crop) and
`fitting`: the second half of BASELINE's metric, frames/sec of the pose-fitting loops
(configs[2..4]) on FIXED workloads at every N: `single_1` / `single_12` one
optimisation step of fitting_single (C3: 196 rays x 192 shared depths, both fields, forward +
losses + backward into the six pose leaves + Adam; the pipelined step of fit_frame;
`single_12_autograd` the same step through autograd, `single_12_dense` with the far-field skip
off), `frames_sharded_12` (C4: 8 frames x 200 steps dealt to the ranks by fit_frames_sharded, no
data-path collective: strong scaling), `video_1234` (C5: ONE 32-frame sequence through
fit_sequence_video -- 29 windows x 5 passes, windows sharded over the ranks, one all-reduce of
the pose-gradient block per step), `video_1234_step` (one window step), `video_1234_weak` (a
3 + 2 x world-frame sequence).  `fitting.roofline` prices the step on the samples it EXECUTED
(the compacted launches' live counts), `roofline_dense` the dense step on the dense count.

`training` (rank 0): one iteration of exp_runner.train's inner loop per field kind.

`--gpus N` without a torch.distributed environment starts the N ranks itself
(`python -m torch.distributed.run`, as a child process, before anything touches the
GPU) and relays rank 0's line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

H_IMG = W_IMG = 512
N_SAMPLES = 64
NEAR, FAR = 0.4, 1.5
# algorithmic work per ray-sample of the hand field: 1 sdf forward + 1 input-gradient
# sweep + 1 colour forward = 2 * 1,234,176 + 624,640 MAC = 6.186 MFLOP
HAND_FLOP_PER_SAMPLE = 2.0 * (2 * 1234176 + 624640)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: bf16/f16 MFMA, dense


def build_scene(dev, seed, precision='f16x3'):
    from honerf_amd import synth
    from honerf_amd.nets import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
    from honerf_amd.renderer import NeuSRenderer
    sdf, col, var = SDFNetwork().to(dev), RenderingNetwork(use_gradients=True).to(dev), SingleVarianceNetwork(0.3).to(dev)
    sdf.reset_parameters(21)
    col.reset_parameters(22)
    ren = NeuSRenderer(sdf, var, col, 'hand', N_SAMPLES, 0, 0, 4, 1.0)
    ren.precision = precision
    bt_inv, T_pose, joints = synth.synth_hand_pose(seed)
    cam = synth.front_camera(dist=0.0, focal=2.0)
    # the hand (~0.2 m across at z ~ 0.95) fills about half of the image width
    xy = synth.ndc_grid(H_IMG, W_IMG) * 0.45
    xy[:, 1] += 0.12
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    scene = dict(xy=t(xy), R=t(cam['R']), T=t(cam['T']), focal=t(cam['focal']), principal=t(cam['principal']),
                 bt_inv=t(bt_inv), T_pose=t(T_pose), t_rand=torch.rand(H_IMG * W_IMG, 1, device=dev,
                                                                         generator=torch.Generator(dev).manual_seed(1)))
    return ren, sdf, col, scene


def cpu_baseline(sdf, col, scene, crop=96, threads=32):
    """The CPU oracle (port of the reference's PyTorch path) on a crop x crop centre block of
    the same frame, on `threads` host cores (more than ~32 only adds contention at these sizes)."""
    from oracle.nets import Field
    from oracle import render as orr
    cores = min(threads, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cpu = lambda v: v.detach().cpu()
    field = Field('hand', {k: cpu(v) for k, v in sdf.state_dict().items()},
                  {k: cpu(v) for k, v in col.state_dict().items()}, 0.3)
    idx = torch.arange(H_IMG * W_IMG).reshape(H_IMG, W_IMG)
    a = (H_IMG - crop) // 2
    sel = idx[a:a + crop, a:a + crop].reshape(-1)
    xy = cpu(scene['xy'])[sel]
    o, d = orr.rays_from_xy(xy, cpu(scene['R'])[0], cpu(scene['T'])[0], cpu(scene['focal'])[0], cpu(scene['principal'])[0])
    tr = cpu(scene['t_rand'])[sel]
    t0 = time.perf_counter()
    n_done = 0
    chunk = 400
    for s in range(0, o.shape[0], chunk):
        orr.render_single(field, o[s:s + chunk], d[s:s + chunk], NEAR, FAR, tr[s:s + chunk], N_SAMPLES, 0, 4,
                          bt_inv=cpu(scene['bt_inv']), T_pose=cpu(scene['T_pose']))
        n_done += o[s:s + chunk].shape[0]
    dt = time.perf_counter() - t0
    return {'value': n_done * N_SAMPLES / dt, 'unit': 'ray-samples/s', 'cores': cores, 'kind': 'port',
            'sample': '%dx%d centre crop of the same 512x512x64 frame (%d ray-samples, %.1f s)'
                      % (crop, crop, n_done * N_SAMPLES, dt)}


# ---- C1 (BASELINE configs[0]): the reference's own CPU-runnable case ------------------------------------
C1_H = C1_W = 128
C1_SAMPLES = 32


def build_scene_c1(dev, precision='f16x3'):
    """`exp_runner.py --mode test` on the object conf: obj nets at conf size, one view 128 x 128, 32 samples per ray
    (n_importance = 0), camera R = I, T = (0,0,1), f = 2, Ro = I, To = 0 (exp_runner.py:336-367)."""
    from honerf_amd import synth
    from honerf_amd.nets import SDFNetwork_OBJ, RenderingNetwork_OBJ, SingleVarianceNetwork
    from honerf_amd.renderer import NeuSRenderer
    sdf, col, var = SDFNetwork_OBJ().to(dev), RenderingNetwork_OBJ().to(dev), SingleVarianceNetwork(0.3).to(dev)
    sdf.reset_parameters(11)
    col.reset_parameters(12)
    ren = NeuSRenderer(sdf, var, col, 'obj', C1_SAMPLES, 0, 0, 4, 1.0)
    ren.precision = precision
    cam = synth.front_camera(dist=1.0, focal=2.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B = C1_H * C1_W
    sc = dict(xy=t(synth.ndc_grid(C1_H, C1_W)), R=t(cam['R']), T=t(cam['T']), focal=t(cam['focal']), principal=t(cam['principal']),
              Ro=torch.eye(3, device=dev), To=torch.zeros(3, device=dev),
              t_rand=torch.rand(B, 1, generator=torch.Generator('cpu').manual_seed(1)).to(dev))
    return ren, sdf, col, sc


def c1_oracle(sdf, col, sc, sel=None, threads=32, chunk=441):
    """The CPU oracle on C1 (rows `sel` of the frame, default all), in the reference's chunks of 441 rays
    (exp_runner.py:356-367) -> (colours, seconds)."""
    from oracle.nets import Field
    from oracle import render as orr
    torch.set_num_threads(min(threads, os.cpu_count() or 1))
    cpu = lambda v: v.detach().cpu()
    field = Field('obj', {k: cpu(v) for k, v in sdf.state_dict().items()}, {k: cpu(v) for k, v in col.state_dict().items()}, 0.3)
    xy, tr = cpu(sc['xy']), cpu(sc['t_rand'])
    if sel is not None:
        xy, tr = xy[sel], tr[sel]
    o, d = orr.rays_from_xy(xy, cpu(sc['R'])[0], cpu(sc['T'])[0], cpu(sc['focal'])[0], cpu(sc['principal'])[0])
    t0 = time.perf_counter()
    outs = [orr.render_single(field, o[s:s + chunk], d[s:s + chunk], NEAR, FAR, tr[s:s + chunk], C1_SAMPLES, 0, 4,
                              Ro=cpu(sc['Ro']), To=cpu(sc['To']))['color_fine'].detach() for s in range(0, o.shape[0], chunk)]
    return torch.cat(outs), time.perf_counter() - t0


def render_c1(ren, sc, lib_mod):
    lib = lib_mod.load()
    B = sc['xy'].shape[0]
    o, d = torch.empty(B, 3, device=sc['xy'].device), torch.empty(B, 3, device=sc['xy'].device)
    lib_mod.check(lib.hn_ray_gen(lib_mod.ptr(sc['xy']), lib_mod.ptr(sc['R']), lib_mod.ptr(sc['T']), lib_mod.ptr(sc['focal']),
                                 lib_mod.ptr(sc['principal']), 1, B, lib_mod.ptr(o), lib_mod.ptr(d), lib_mod.stream_ptr()), 'hn_ray_gen')
    with torch.no_grad():      # `--mode test`: no graph (NeuSRenderer.render is differentiable when the caller can differentiate it)
        return ren.render(o, d, NEAR, FAR, None, None, None, sc['Ro'], sc['To'], 0, t_rand=sc['t_rand'])


def time_c1(dev, precision, with_cpu):
    """C1 in full on the GPU (ms per frame) and, with_cpu, on the host cores through the oracle (the reference's own
    CPU-runnable configuration: BASELINE.md section 3)."""
    from honerf_amd import lib as L
    ren, sdf, col, sc = build_scene_c1(dev, precision)
    out = render_c1(ren, sc, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = render_c1(ren, sc, L)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    n = C1_H * C1_W * C1_SAMPLES
    res = {'workload': 'C1: obj nets (conf size), 128x128 rays x 32 samples, n_importance=0, one view', 'ms_per_frame': sec * 1e3,
           'value': n / sec, 'unit': 'ray-samples/s', 'weight_sum_mean': float(out['weight_sum'].mean())}
    if with_cpu:
        ref, dt = c1_oracle(sdf, col, sc)
        err = float((out['color_fine'].cpu() - ref).abs().max() / ref.abs().max())
        res['cpu_baseline'] = {'value': n / dt, 'unit': 'ray-samples/s', 'cores': min(32, os.cpu_count() or 1), 'kind': 'port',
                               'sample': 'the whole C1 frame (%d ray-samples, %.1f s), chunks of 441 rays' % (n, dt)}
        res['colour_rel_err_vs_cpu'] = err
    return res


# ---- the fitting loops (BASELINE configs[2..4]) --------------------------------------------------------------
FIT_RAYS, FIT_N, FIT_IMP = 196, 64, 64            # fit_confs/fit_1_8views.conf:24, 86-92 -> 192 shared depths
VID_FRAMES, VID_RAYS = 4, 40                      # fitting_video.py:146-149, 264-266
STEPS_PER_FRAME = {'1': 30 * 8, '12': 25 * 8}     # fitting_single.py:124-132, 200-201
OBJ_FLOP_PER_SAMPLE = 2.0 * (2 * 524544 + 292864)
# algorithmic work of one step: the final evaluation of both fields (1 sdf forward + 1 input-gradient sweep + 1 colour
# forward each) + its adjoint (colour backward + forward-direction sweep + second reverse sweep = the same product
# count again, + the 257-row W8) + the sdf-only evaluations of the importance rounds (64 + 3 x 16 per ray and field)
HAND_SDF_FLOP, OBJ_SDF_FLOP = 2.0 * 1234176, 2.0 * 524544


def fit_step_flop(n_rays, S=FIT_N + 2 * FIT_IMP, hand_final=None, hand_coarse=None):
    """Algorithmic FLOP of one fitting_single step.  hand_final / hand_coarse: the samples the hand field EXECUTES in the final
    evaluation (+ its adjoint) and in the coarse sdf pass when the exact far-field skip is on (None: every sample, "dense")."""
    n_final = n_rays * S
    hf = n_final if hand_final is None else hand_final
    hc = n_rays * FIT_N if hand_coarse is None else hand_coarse
    final = hf * HAND_FLOP_PER_SAMPLE + n_final * OBJ_FLOP_PER_SAMPLE
    adjoint = hf * (HAND_FLOP_PER_SAMPLE + 2.0 * 257 * 256) + n_final * (OBJ_FLOP_PER_SAMPLE + 2.0 * 257 * 256)
    sampling = (hc + n_rays * 3 * (FIT_IMP // 4)) * HAND_SDF_FLOP + n_rays * (FIT_N + 3 * (FIT_IMP // 4)) * OBJ_SDF_FLOP
    return final + adjoint + sampling


def executed_hand_samples(ren, n_rays):
    """(final evaluation, coarse pass): the hand samples the LAST differentiable render of `ren` evaluated, read from the compaction
    records it left on the device (hn_render_dual_compact_offsets); None where that launch ran dense."""
    import ctypes
    from honerf_amd import lib as L
    lib = L.load()
    hand, obj = ren.fields()
    offs = (ctypes.c_size_t * 2)()
    L.check(lib.hn_render_dual_compact_offsets(hand.handle, obj.handle, n_rays, ren.n_samples, ren.n_importance, ren.up_sample_steps, offs),
            'hn_render_dual_compact_offsets')
    none = ctypes.c_size_t(-1).value
    out = []
    for off, buf in ((offs[0], ren._tape.buf), (offs[1], ren._ws.buf)):
        out.append(None if (off == none or buf is None) else int(buf[off:off + 4].view(torch.int32).item()))
    return tuple(out)


def build_fit_nets(dev, n_frames, precision):
    """Both fields at conf size behind the (frame-batched for n_frames > 1) two-field renderer."""
    from honerf_amd.nets import (SDFNetwork, RenderingNetwork, SDFNetwork_OBJ, RenderingNetwork_OBJ, SingleVarianceNetwork)
    from honerf_amd.renderer import NeuSRenderer_fitting
    from honerf_amd.renderer_batch import NeuSRenderer_fitting as Batched
    nets = [SDFNetwork(use_batch=n_frames > 1), SingleVarianceNetwork(0.3), RenderingNetwork(use_gradients=True),
            SDFNetwork_OBJ(), SingleVarianceNetwork(0.3), RenderingNetwork_OBJ()]
    for m, sd in zip((nets[0], nets[2], nets[3], nets[5]), (21, 22, 11, 12)):
        m.reset_parameters(sd)
    nets = [m.to(dev) for m in nets]
    ren = (Batched if n_frames > 1 else NeuSRenderer_fitting)(*nets, FIT_N, FIT_IMP, 0, 4, 1.0)
    ren.precision = precision
    return ren, nets


def build_fit_data(dev, seed, n_frames, halo=True, drift=0.0, obj_offset=(0.02, 0.0, 0.01)):
    """A synthetic frame (or sequence of n_frames: the same pose with a small per-frame drift) and its pose chain:
    halo = the reference's six refine leaves through hn_pose_chain (fitting_single.py:183-226, honerf_amd.fitting.HaloPoseChain),
    else the reduced rigid chain (palm + object motion only).  obj_offset: the object's centre relative to joint 9 (the synthetic
    object field is a sphere-like SDF of radius ~0.4 in its own frame: the default puts the hand deep inside it -- penetration
    everywhere, no contact; the whole-step tests' second scene moves the centre so that its SURFACE crosses the hand)."""
    from honerf_amd import fitting as F, synth
    bt, tp, j = synth.synth_hand_pose(seed)
    R, tt = synth.synth_obj_pose(seed + 1, center=tuple(j[9] + np.asarray(obj_offset, dtype=np.float64)))
    rng = np.random.RandomState(seed)
    u = rng.standard_normal((2000, 3))
    verts = (u / np.linalg.norm(u, axis=1, keepdims=True) * 0.025).astype(np.float32)
    rep = lambda a: np.repeat(a[None], n_frames, 0)
    jj, tts = rep(j), rep(tt)
    if drift:
        steps = np.cumsum(rng.standard_normal((n_frames, 1, 3)).astype(np.float32) * drift, axis=0)
        jj, tts = jj + steps, tts + steps[:, 0]
    if halo:
        chain = F.HaloPoseChain(jj, F.bone_lengths_of(jj), None, rep(R), tts, verts, device=dev)
    else:
        chain = F.RigidPoseChain(rep(bt), rep(tp), jj, rep(R), tts, verts, device=dev)
    return chain, j, torch.from_numpy(verts).to(dev)


def build_fit(dev, seed, n_frames, rays, precision, halo=False, obj_offset=(0.02, 0.0, 0.01)):
    from honerf_amd import fitting as F
    ren, nets = build_fit_nets(dev, n_frames, precision)
    chain, j, verts = build_fit_data(dev, seed, n_frames, halo, obj_offset=obj_offset)
    views = F.synthetic_views(8, n_frames, rays, seed, j[9], device=dev)
    return ren, nets, chain, views, verts[None].expand(n_frames, -1, -1).contiguous()


C4_FRAMES = 8        # BASELINE configs[3]: the 8-frame batch of fitting_single fit_12_8views, the SAME eight frames at every N
C5_FRAMES = 32       # a 32-frame sequence, 29 sliding windows, 5 passes, the SAME sequence at every N


def time_fit(dev, dist, rank, world, precision, steps, warmup, outer_iters=5, windows_per_rank=2, quick=False):
    """The fitting half of BASELINE's metric.  Per-step times (every rank its own frame / window, no collective, max over
    ranks) on the reference's six-leaf pose chain, then the two sharded loops as the product runs them, on workloads that do
    not depend on the number of GPUs (strong scaling: the N = 1, 2, 4, 8 lines fit the same frames):
    C4 `fit_frames_sharded`: the same C4_FRAMES fit_12 frames dealt out over the ranks (all 25 x 8 steps of each);
    C5 `fit_sequence_video`: the same C5_FRAMES-frame sequence, `outer_iters` passes over its 29 windows, window-parallel with
    the pose-gradient all-reduce per step.  (`video_1234_weak`: the round-3 variant, 3 + windows_per_rank x world frames.)"""
    from honerf_amd import fitting as F
    from honerf_amd import lib as _L
    res = {}
    _L.dropped_samples(reset=True)

    def wall(fn):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, out

    WINDOWS = 3

    def timed(fn):
        """Seconds per step of a per-step leg: `warmup` untimed steps, then WINDOWS timed windows of `steps` steps each, the MEDIAN window.
        (As timeit does, what earlier legs left behind is collected before the leg and the cyclic collector stays off inside the windows.  The
        median of three: a window is 0.2 s, and on a box whose host was busy with something else the one leg that goes through autograd --
        1.4 ms of host work per 2.3 ms window step -- read 3.2 and 4.3 ms in a single window beside 2.3 in the sequence leg of the same
        process; the windows are kept in `timed.last`.)"""
        import gc
        gc.collect()
        torch.cuda.synchronize()
        for i in range(warmup):
            fn(i)
        was = gc.isenabled()
        gc.disable()
        try:
            secs = [wall(lambda: [fn(warmup + w * steps + i) for i in range(steps)])[0] / steps for w in range(WINDOWS)]
        finally:
            if was:
                gc.enable()
        timed.last = [round(x * 1e3, 4) for x in secs]
        return sorted(secs)[len(secs) // 2]

    # ---- C3 / C4 per step -----------------------------------------------------------------------------------------
    ren, nets = build_fit_nets(dev, 1, precision)
    chain, j, verts = build_fit_data(dev, 40 + rank, 1, halo=True)
    views = F.synthetic_views(8, 1, FIT_RAYS, 40 + rank, j[9], device=dev)
    opt = F.make_optimizer(chain, video=False)
    for ft in ('1', '12'):
        # the step as fit_frame runs it: PipelinedSingleFit (the hand's and the object's halves on two streams that stay apart across
        # steps); `single_12_autograd`: the same step through autograd on one stream + the library's fork / join inside the render
        sec = timed(lambda i: F.fit_step(ren, views[i % 8], chain, opt, NEAR, FAR, ft, pipelined=True))
        res['single_' + ft] = {'ms_per_step': sec * 1e3, 'ms_per_step_timed_windows': list(timed.last), 'steps_per_frame': STEPS_PER_FRAME[ft],
                               'frames_per_s': world / (STEPS_PER_FRAME[ft] * sec), 'pose_chain': 'halo (six refine leaves, hn_pose_chain)',
                               'form': 'PipelinedSingleFit (two streams across steps)'}
    F.finish_pipeline(opt)
    sec = timed(lambda i: F.fit_step(ren, views[i % 8], chain, opt, NEAR, FAR, '12'))
    res['single_12_autograd'] = {'ms_per_step': sec * 1e3, 'what': 'secondary: fit_backward + fit_apply through autograd (what round 3 timed)'}
    # what the hand field executed in those steps (the exact far-field skip): mean over the 8 views of the live-sample counts
    live = []
    for v in range(8):
        F.fit_backward(ren, views[v], chain, NEAR, FAR, '12')
        torch.cuda.synchronize()
        live.append(executed_hand_samples(ren, FIT_RAYS))
    mean_of = lambda k: (None if any(x[k] is None for x in live) else float(np.mean([x[k] for x in live])))
    res['single_12']['hand_samples_executed'] = {'final_evaluation': mean_of(0), 'coarse_pass': mean_of(1),
                                                  'of': [FIT_RAYS * (FIT_N + 2 * FIT_IMP), FIT_RAYS * FIT_N],
                                                  'what': 'mean over the 8 views of the device-side counts of the compacted launches (live samples + 1)'}
    # the same step with every sample of the hand field evaluated ("dense": NeuSRenderer_fitting.compact_far_field = False; the
    # default skips the samples whose bone masks are all exactly 0 -- bit-identical outputs)
    ren.compact_far_field = False
    sec = timed(lambda i: F.fit_step(ren, views[i % 8], chain, opt, NEAR, FAR, '12', pipelined=True))
    F.finish_pipeline(opt)
    res['single_12_dense'] = {'ms_per_step': sec * 1e3, 'what': 'secondary: no far-field skip (every one of the 196 x 192 samples through the hand field)'}
    ren.compact_far_field = True
    chain_r, _, _ = build_fit_data(dev, 40 + rank, 1, halo=False)
    opt_r = F.make_optimizer(chain_r, video=False)
    sec = timed(lambda i: F.fit_step(ren, views[i % 8], chain_r, opt_r, NEAR, FAR, '12'))
    res['single_12_rigid_chain'] = {'ms_per_step': sec * 1e3, 'what': 'secondary: palm + object motion only (RigidPoseChain)'}
    single = (ren, nets, chain_r, views)

    # ---- C4: the same C4_FRAMES fit_12 frames at every N, dealt out over the ranks, the whole 25 x 8-step loop of every frame ------
    n_c4 = 2 if quick else C4_FRAMES

    def make_frame(f):
        ch, jf, _ = build_fit_data(dev, 140 + f, 1, halo=True)
        return F.synthetic_views(8, 1, FIT_RAYS, 140 + f, jf[9], device=dev), ch
    def c4_leg(batch, n_iters=None):
        made = {f: make_frame(f) for f in F.shard_frames(n_c4, rank, world)}           # data "loading" is not timed
        def frame(f):
            torch.manual_seed(5000 + f)      # the frame's seed, set where its data is handed over: its jitter stream (per-frame seeds)
            return made[f]
        dt, red = wall(lambda: F.fit_frames_sharded(ren, n_c4, frame, NEAR, FAR, '12', n_iters=n_iters, dist=dist, batch=batch))
        return {'frames': red['frames'], 'steps_per_frame': STEPS_PER_FRAME['12'], 'seconds': dt,
                'frames_per_s': red['frames'] / dt, 'ms_per_step': dt / (STEPS_PER_FRAME['12'] * max(len(made), 1)) * 1e3,
                'frames_per_rank': len(made), 'frame_batch': min(batch, max(len(made), 1)), 'scaling': 'strong (the same %d frames at every N)' % n_c4,
                'collectives': 'one SUM all-reduce of %d loss values at the end' % len(F.LOSS_KEYS), 'loss_mean': red['loss']}
    # a rank that owns several frames fits up to FRAME_BATCH of them side by side through the same launches (per-frame results equal
    # the one-by-one fits to the bit: tests/test_gpu_surface.py); `frames_sharded_12_one_by_one`: the same frames strictly one after the other
    # (untimed: one pass of 8 steps at the batched sizes -- the renderer's grow-only workspaces and the 2.6 GB-per-frame tape are
    # allocated once per process and re-used by every later batch; the one-by-one sizes were warmed by the step legs above)
    c4_leg(F.FRAME_BATCH, n_iters=1)
    res['frames_sharded_12'] = c4_leg(F.FRAME_BATCH)
    res['frames_sharded_12']['what'] = ('C4: fitting_single fit_12_8views, %d frames dealt out over the GPUs, fit_frames_sharded end to end; a rank fits up to %d of '
                                        'its frames side by side (fit_frames_batched)' % (n_c4, F.FRAME_BATCH))
    res['frames_sharded_12_one_by_one'] = c4_leg(1)
    res['frames_sharded_12_one_by_one']['what'] = 'secondary: the same frames, every rank fitting its frames strictly one after the other (frame batch 1: what round 4 timed)'
    res['frames_sharded_12']['loss_mean_equals_one_by_one'] = res['frames_sharded_12']['loss_mean'] == res['frames_sharded_12_one_by_one']['loss_mean']

    # ---- C5 per step (one window per rank, no collective) ---------------------------------------------------------
    renb, netsb = build_fit_nets(dev, VID_FRAMES, precision)

    def sequence(data_num):
        chain_s, j_s, v_s = build_fit_data(dev, 60, data_num, halo=True, drift=0.002)      # the SAME sequence on every rank
        return chain_s, j_s, v_s[None].expand(VID_FRAMES, -1, -1).contiguous()
    data_num = 3 + windows_per_rank * world
    chainb, jb, ov = sequence(data_num)
    optb = F.make_optimizer(chainb, video=True)
    wins = F.sliding_windows(data_num)
    my = wins[rank % len(wins)]
    vwin = F.synthetic_views(8, VID_FRAMES, VID_RAYS, 60 + rank, jb[9], device=dev)
    sec_v = timed(lambda i: F.fit_step(renb, vwin[i % 8], chainb, optb, NEAR, FAR, '1234', index=my, smooth_ends=(my[0] == 0, False),
                                       obj_verts_for_stable=ov))
    res['video_1234_step'] = {'ms_per_step': sec_v * 1e3, 'ms_per_step_timed_windows': list(timed.last), 'steps_per_window': 32, 'windows_per_s': world / (32 * sec_v),
                              'what': 'one window per rank, no collective (the N = 1 step)'}

    # ---- C5: the sequence loop, window-parallel, pose-gradient all-reduce between backward and Adam --------------------
    def run_sequence(n_frames, passes, what):
        chains, js, ovs = sequence(n_frames)
        per_window = {tuple(w): F.synthetic_views(8, VID_FRAMES, VID_RAYS, 300 + w[0], js[9], device=dev)
                      for w in F.window_schedule(n_frames, rank, world) if w is not None}          # data "loading" is not timed

        def window_views(index, vid, step):
            return per_window[tuple(index)][vid]
        window_views.n_views = 8
        dt, st = wall(lambda: F.fit_sequence_video(renb, window_views, chains, NEAR, FAR, n_frames, '1234', outer_iters=passes, obj_verts=ovs,
                                                   dist=dist))
        n_win = len(F.sliding_windows(n_frames))
        rounds = (n_win + world - 1) // world
        return {'ideal_rounds_cap': n_win / float(rounds * world),
                'ideal_rounds_cap_what': '%d windows in %d rounds of %d concurrent windows: the efficiency this schedule can reach before any latency'
                                         % (n_win, rounds, world),
                'ms_per_step': dt / st['steps'] * 1e3, 'steps': st['steps'], 'data_num': n_frames, 'windows': n_win, 'outer_iters': passes,
                'seconds': dt, 'windows_per_s': n_win * passes / dt, 'frames_per_s': n_frames / dt,
                'allreduce_calls': st['allreduce_calls'], 'allreduce_floats_per_step': st['allreduce_floats'] // max(st['allreduce_calls'], 1),
                'efficiency_vs_unsynced_step': sec_v / (dt / st['steps']), 'pose_chain': 'halo (six refine leaves, hn_pose_chain)', 'what': what}
    n_c5 = 8 if quick else C5_FRAMES
    res['video_1234'] = run_sequence(n_c5, 1 if quick else outer_iters,
                                     'C5: fitting_video fit_1234_8views over the SAME %d-frame sequence at every N (strong scaling over its %d windows; rounds of '
                                     '`world` concurrent windows), fit_sequence_video: one SUM all-reduce of the data_num x 45 pose-gradient block per step'
                                     % (n_c5, n_c5 - 3))
    res['video_1234_weak'] = run_sequence(data_num, 1 if quick else outer_iters,
                                          'secondary (round 3\'s workload): a %d-frame sequence = %d windows per rank and pass' % (data_num, windows_per_rank))
    # samples the hand adjoint dropped over ALL fitting legs of this rank (out of the fp16 fragments' range next to a bone's origin: g_pts = 0,
    # no share in the pose gradients; hn_dropped_samples) -- 0 on these scenes means no step's gradient was touched by the rule
    res['dropped_samples'] = _L.dropped_samples()
    return res, single


def cpu_fit_baseline(single, rays=49, threads=32):
    """One fitting_single step of the CPU oracle (autograd through oracle.render.render_dual + the same losses) on
    `rays` of the 196 rays (the step's cost is linear in the ray count)."""
    from oracle.nets import Field
    from oracle import render as orr
    from honerf_amd import fitting as F
    ren, nets, chain, views = single
    cores = min(threads, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cpu = lambda v: v.detach().cpu()
    sd = lambda m: {k: cpu(v) for k, v in m.state_dict().items()}
    hand = Field('hand', sd(nets[0]), sd(nets[2]), 0.3)
    obj = Field('obj', sd(nets[3]), sd(nets[5]), 0.3)
    v = views[0]
    pose = {k: cpu(x) for k, x in chain().items()}
    bt = pose['bt_inv'][0].clone().requires_grad_(True)
    Ro = pose['obj_r'][0].T.contiguous().clone().requires_grad_(True)
    To = pose['obj_t'][0].clone().requires_grad_(True)
    o, d = orr.rays_from_xy(cpu(v['xy'])[:rays], cpu(v['cam']['R'])[0], cpu(v['cam']['T'])[0], cpu(v['cam']['focal'])[0],
                            cpu(v['cam']['principal'])[0])
    t0 = time.perf_counter()
    out = orr.render_dual(hand, obj, o, d, NEAR, FAR, torch.rand(rays, 1), FIT_N, FIT_IMP, 4, bt, pose['T_pose_21'][0], Ro, To)
    terms = F.render_loss_terms(out, cpu(v['true_rgb'])[:rays], cpu(v['true_mask'])[:rays], '12')
    terms['loss'].backward()
    dt = time.perf_counter() - t0
    return {'value': 1.0 / (dt * FIT_RAYS / rays), 'unit': 'fitting_single steps/s (196 rays)', 'cores': cores, 'kind': 'port',
            'sample': 'one step (render + losses + backward) on %d of the 196 rays: %.1f s, scaled by 196/%d' % (rays, dt, rays)}


def pmc_traffic(kernel):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC summary of this same workload
    (profiles/r*/pmc_bench_*.json: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot collect counters itself; the newest
    summary for this kernel is quoted ONLY IF it was collected on the kernel sources of this tree (tools/pmc_summary.py stamps
    their hash, tools/srchash.py): a summary of other sources is reported as stale, not as `traffic`.
    -> (bytes or None, source path or None, note or None)"""
    import glob
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    now = srchash.source_hash()
    newest = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'pmc_bench_*.json'))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if (kernel in str(d.get('kernel')) or kernel.replace('<1>', '<true>') in str(d.get('kernel'))) and 'derived' in d and 'hbm_bytes_per_launch' in d['derived']:
            newest = (d['derived']['hbm_bytes_per_launch'], os.path.relpath(path, ROOT), d.get('csrc_sha16'))
    if newest is None:
        return None, None, 'no PMC summary of this kernel is committed'
    if newest[2] != now:
        return None, newest[1], ('STALE: %s was collected on kernel sources %s, this tree is %s (%.4g bytes per launch there); re-run tools/profile_bench.sh'
                                 % (newest[1], newest[2] or 'unstamped', now, newest[0]))
    return newest[0], newest[1], 'collected on the kernel sources of this tree (csrc_sha16 %s)' % now


def fit_kernel_traffic():
    """HBM-side bytes per fitting_single step of the four field kernels of a step (taped evaluation + adjoint of both fields), from the
    committed PMC summaries of tools/profile_fit.sh (profiles/r*/pmc_fit_k_field2_{hand3,hand4,obj3,obj4}.json, one frame per launch) --
    quoted only if they were collected on the kernel sources of this tree.  -> (bytes or None, note)"""
    import glob
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    now = srchash.source_hash()
    dirs = sorted(d for d in glob.glob(os.path.join(ROOT, 'profiles', 'r*')) if os.path.exists(os.path.join(d, 'pmc_fit_k_field2_hand3.json')))
    if not dirs:
        return None, 'no PMC summary of the fitting kernels is committed'
    total, parts = 0.0, {}
    for k in ('hand3', 'hand4', 'obj3', 'obj4'):
        try:
            d = json.load(open(os.path.join(dirs[-1], 'pmc_fit_k_field2_%s.json' % k)))
        except Exception:
            return None, 'incomplete PMC summaries in %s' % os.path.relpath(dirs[-1], ROOT)
        if d.get('csrc_sha16') != now:
            return None, 'STALE: %s was collected on kernel sources %s, this tree is %s' % (os.path.relpath(dirs[-1], ROOT), d.get('csrc_sha16'), now)
        parts[k] = d['derived']['hbm_bytes_per_launch']
        total += parts[k]
    return total, ('k_field2_hand<3> %.3g + hand<4> %.3g + obj<3> %.3g + obj<4> %.3g bytes per one-frame step (tapes written and read back: ~5.4 + 3.4 MB per 128-sample '
                   'tile), %s, csrc_sha16 %s' % (parts['hand3'], parts['hand4'], parts['obj3'], parts['obj4'], os.path.relpath(dirs[-1], ROOT), now))


def spawn_command(n_gpus, argv):
    """The launch line of the driver contract: one rank per GPU of one node over RCCL, rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def rccl_one_rank_leg(quick=False):
    """RCCL on the hardware there is: a process group of ONE rank over the `nccl` backend (= RCCL) on this GPU, created before
    anything else touches the device, and the two sharded loops driven through it with their collectives NOT short-circuited at
    world == 1 (fitting.FORCE_COLLECTIVE): `fit_sequence_video` issues one device-side all-reduce of the pose-gradient block per step
    between the backward pass and Adam, `fit_frames_sharded` the reduction of the loss vector.  A one-rank SUM is the identity, so
    the run must equal the run without a collective TO THE BIT; it also says what the collective costs a step in launch latency.
    Runs as a child process of `bench.py --gpus 1` (and of tests/test_gpu_surface.py); prints one JSON line."""
    import datetime
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from honerf_amd.pose import bind_streams
    bind_streams(dev)
    import torch.distributed as dist
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(minutes=5))
        probe = torch.full((4,), 3.0, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert probe.tolist() == [3.0] * 4
    finally:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    from honerf_amd import fitting as F
    res = {'backend': dist.get_backend(), 'world': dist.get_world_size(), 'rccl_version': list(torch.cuda.nccl.version()),
           'device': torch.cuda.get_device_name(0)}
    n_frames = 6 if quick else 8
    renb, _ = build_fit_nets(dev, VID_FRAMES, 'f16x3')

    def run_video(force):
        F.FORCE_COLLECTIVE = force
        torch.manual_seed(77)
        chain, j, v = build_fit_data(dev, 60, n_frames, halo=True, drift=0.002)
        ov = v[None].expand(VID_FRAMES, -1, -1).contiguous()
        per_window = {tuple(w): F.synthetic_views(2 if quick else 8, VID_FRAMES, VID_RAYS, 300 + w[0], j[9], device=dev) for w in F.sliding_windows(n_frames)}

        def window_views(index, vid, step):
            return per_window[tuple(index)][vid]
        window_views.n_views = 2 if quick else 8
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = F.fit_sequence_video(renb, window_views, chain, NEAR, FAR, n_frames, '1234', outer_iters=1, obj_verts=ov, dist=dist)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return [p.detach().clone() for p in chain.parameters()], st, dt
    pw, _, _ = run_video(False)                           # warm (allocations, first-use costs)
    p0, st0, dt0 = run_video(False)
    p1, st1, dt1 = run_video(True)
    maxdiff = lambda a, b: max(float((x - y).abs().max()) for x, y in zip(a, b))
    res['diag'] = {'unforced_run_equals_its_repeat': all(torch.equal(a, b) for a, b in zip(pw, p0)), 'unforced_vs_repeat_maxdiff': maxdiff(pw, p0),
                   'forced_vs_unforced_maxdiff': maxdiff(p0, p1)}
    if os.environ.get('HONERF_RCCL_DIAG') == '1':
        p1b, _, _ = run_video(True)
        res['diag']['forced_run_equals_its_repeat'] = all(torch.equal(a, b) for a, b in zip(p1, p1b))
        res['diag']['forced_vs_repeat_maxdiff'] = maxdiff(p1, p1b)
    res['fit_sequence_video'] = {
        'data_num': n_frames, 'steps': st1['steps'], 'allreduce_calls': st1['allreduce_calls'],
        'allreduce_calls_equal_steps': st1['allreduce_calls'] == st1['steps'],
        'allreduce_floats_per_step': st1['allreduce_floats'] // max(st1['allreduce_calls'], 1),
        'allreduce_calls_without_force': st0['allreduce_calls'],
        'ms_per_step_with_collective': dt1 / st1['steps'] * 1e3, 'ms_per_step_without': dt0 / st0['steps'] * 1e3,
        'bit_identical_to_the_run_without_collective': all(torch.equal(a, b) for a, b in zip(p0, p1)),
        'what': 'one pass over the %d windows of a %d-frame sequence, fit type 1234; the SUM all-reduce of the [data_num x 45] gradient block in place, '
                'between fit_backward and the Adam step' % (n_frames - 3, n_frames)}
    ren1, _ = build_fit_nets(dev, 1, 'f16x3')

    def run_frames(force):
        F.FORCE_COLLECTIVE = force
        made = {}
        for f in range(2):
            torch.manual_seed(500 + f)
            ch, jf, _ = build_fit_data(dev, 140 + f, 1, halo=True)
            made[f] = (F.synthetic_views(8, 1, FIT_RAYS, 140 + f, jf[9], device=dev), ch)
        torch.manual_seed(78)
        out = F.fit_frames_sharded(ren1, 2, lambda f: made[f], NEAR, FAR, '12', n_iters=2 if quick else 5, dist=dist)
        torch.cuda.synchronize()
        return out, [p.detach().clone() for f in range(2) for p in made[f][1].parameters()]
    o0, q0 = run_frames(False)
    o1, q1 = run_frames(True)
    res['fit_frames_sharded'] = {'frames': o1['frames'], 'steps': o1['steps'], 'allreduce_calls': o1['allreduce_calls'],
                                 'allreduce_calls_without_force': o0['allreduce_calls'],
                                 'bit_identical_to_the_run_without_collective': all(torch.equal(a, b) for a, b in zip(q0, q1)) and
                                 all(o0[k] == o1[k] for k in F.LOSS_KEYS)}
    F.FORCE_COLLECTIVE = False
    dist.barrier()
    dist.destroy_process_group()
    res['ok'] = bool(res['fit_sequence_video']['allreduce_calls_equal_steps'] and res['fit_sequence_video']['bit_identical_to_the_run_without_collective']
                     and res['fit_frames_sharded']['allreduce_calls'] == 1 and res['fit_frames_sharded']['bit_identical_to_the_run_without_collective'])
    print(json.dumps(res))
    return 0 if res['ok'] else 4


def spawn_rccl_leg(quick=False, timeout=900):
    """The one-rank RCCL leg as a CHILD process (its process group must come up before anything else touches the GPU, and a failure
    of the communicator must not take the bench line with it) -> its JSON object, or {'error': ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--rccl-one-rank'] + (['--fit-quick'] if quick else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {'error': 'the one-rank RCCL leg did not finish in %d s' % timeout}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith('{'):
            try:
                out = json.loads(line)
                out['returncode'] = r.returncode
                return out
            except ValueError:
                pass
    return {'error': 'no result line (exit %d)' % r.returncode, 'stderr_tail': r.stderr[-1500:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-crop', type=int, default=96)
    ap.add_argument('--no-culled', action='store_true', help='skip the secondary culled measurement')
    ap.add_argument('--precision', default='f16x3', choices=['f16x3', 'fp32'])
    ap.add_argument('--no-f16', action='store_true', help='skip the secondary single-pass f16 measurement')
    ap.add_argument('--no-c1', action='store_true', help='skip the C1 (128x128x32 obj) measurement')
    ap.add_argument('--no-fitting', action='store_true', help='skip the fitting-loop measurements')
    ap.add_argument('--no-training', action='store_true', help='skip the training-iteration measurement')
    ap.add_argument('--fit-steps', type=int, default=80, help='steps per timed window of a per-step fitting leg (10 untimed ones, then 3 windows: the median is reported)')
    ap.add_argument('--fit-outer', type=int, default=5, help='passes over the video sequence (fitting_video.py:157: 5)')
    ap.add_argument('--fit-quick', action='store_true', help='functional check of the fitting legs: 2 frames, an 8-frame sequence, one pass')
    ap.add_argument('--rccl-one-rank', action='store_true', help='run ONLY the one-rank RCCL leg (a process group of one rank over nccl on this GPU, the '
                    'sharded loops with their collectives forced) and print its JSON line')
    ap.add_argument('--no-rccl-leg', action='store_true', help='--gpus 1: skip the one-rank RCCL leg (run as a child process before the measurements)')
    args = ap.parse_args()
    if args.rccl_one_rank:
        sys.exit(rccl_one_rank_leg(args.fit_quick))

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # No torch.distributed environment: start the N ranks ourselves, as a CHILD process, before this process has
        # made any HIP call (a process that has initialised the GPU must never exec or be replaced), and relay its output.
        import subprocess
        sys.exit(subprocess.call(spawn_command(args.gpus, sys.argv[1:])))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node %d\n' % (args.gpus, world, args.gpus))
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    rccl_leg = None
    if world == 1 and not args.no_fitting and not args.no_rccl_leg and args.precision == 'f16x3':
        rccl_leg = spawn_rccl_leg(args.fit_quick)          # a child process, before this process touches the GPU
    # HONERF_BENCH_SHARE_GPU=1: all ranks on device 0 with the gloo backend -- a functional check of the N > 1 code path
    # (sharding, the pose-gradient all-reduce between backward and Adam) on a box with ONE GPU; its timings mean nothing.
    share = os.environ.get('HONERF_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # the library's second stream and the extra stream get their hardware queues before the communicator's stream asks for one
    from honerf_amd.pose import bind_streams
    bind_streams(dev)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        backend = 'gloo' if share else 'nccl'
        # stdout carries ONE JSON line: whatever the communicator libraries print while they connect (gloo does) goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            # an explicit, generous timeout: the fitting legs below keep the ranks apart for tens of seconds between collectives
            if share:
                dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=20))
            else:
                dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(minutes=20))
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                # the first collective builds the communicator: fail here, with a message, not mid-run
            torch.cuda.synchronize()
            assert int(probe.item()) == world
        except Exception as e:
            sys.stderr.write('bench.py: rank %d could not bring up the %s process group of %d ranks on 127.0.0.1 (%s: %s).  RCCL needs one visible GPU per '
                             'rank and HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool; HONERF_BENCH_SHARE_GPU=1 runs the N > 1 code path on ONE GPU over gloo '
                             '(functional check only).\n' % (rank, backend, world, type(e).__name__, e))
            sys.exit(3)
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from honerf_amd import lib as L
    lib = L.load()
    ren, sdf, col, sc = build_scene(dev, seed=9 + rank, precision=args.precision)      # every rank renders its own frame
    B = H_IMG * W_IMG
    rays_o = torch.empty(B, 3, device=dev)
    rays_d = torch.empty(B, 3, device=dev)

    def step():
        L.check(lib.hn_ray_gen(L.ptr(sc['xy']), L.ptr(sc['R']), L.ptr(sc['T']), L.ptr(sc['focal']),
                               L.ptr(sc['principal']), 1, B, L.ptr(rays_o), L.ptr(rays_d), L.stream_ptr()), 'hn_ray_gen')
        with torch.no_grad():      # offline rendering: the plain launch sequence, nothing kept for a backward pass
            return ren.render(rays_o, rays_d, NEAR, FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0,
                              t_rand=sc['t_rand'])

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # the dominant kernel is timed INSIDE the timed steps: the library brackets every field-evaluation launch of these steps with two
    # HIP events on the stream it launches on (hn_debug_field_timer), read back after the region -- kernel time <= step time by construction
    import ctypes
    L.check(lib.hn_debug_field_timer(1), 'hn_debug_field_timer')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.check(lib.hn_debug_field_timer(0), 'hn_debug_field_timer')
    k_total, k_count = ctypes.c_double(0.0), ctypes.c_int(0)
    L.check(lib.hn_debug_field_timer_read(ctypes.byref(k_total), ctypes.byref(k_count)), 'hn_debug_field_timer_read')
    kernel_ms_in_loop = k_total.value / max(k_count.value, 1)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    samples_per_step = B * N_SAMPLES
    value = world * samples_per_step * args.steps / dt

    # ---- the fitting loops: every leg that has a collective runs HERE, on all ranks; after it the process group is torn down and
    #      rank 0 alone takes the secondary measurements (no rank waits in a collective while rank 0 times a CPU oracle)
    fitting, single = None, None
    if not args.no_fitting and args.precision == 'f16x3':
        fitting, single = time_fit(dev, dist, rank, world, args.precision, args.fit_steps, 10, args.fit_outer, quick=args.fit_quick)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    # ---- dominant kernel alone (k_field_hand<true>), HIP events on the launch stream ----------
    field = ren.field()
    z = ren.last_z_vals
    pts = torch.empty(B * N_SAMPLES, 3, device=dev)
    dists = torch.empty(B * N_SAMPLES, device=dev)
    L.check(lib.hn_sample_points(L.ptr(rays_o), L.ptr(rays_d), L.ptr(z), B, N_SAMPLES, 1, (FAR - NEAR) / N_SAMPLES,
                                 L.ptr(pts), L.ptr(dists), L.stream_ptr()), 'hn_sample_points')
    n = B * N_SAMPLES
    o_sdf, o_grad, o_rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    ws_bytes = lib.hn_field_workspace_bytes(field.handle, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    bt, tp = sc['bt_inv'].reshape(1, 21, 4, 4).contiguous(), sc['T_pose'].reshape(1, 21, 3).contiguous()

    def field_launch():
        L.check(lib.hn_field_eval(field.handle, L.ptr(pts), L.ptr(rays_d), n, N_SAMPLES, L.ptr(bt), L.ptr(tp), 1, n,
                                  L.ptr(o_sdf), L.ptr(o_grad), L.ptr(o_rgb), None, L.ptr(ws), ws_bytes, L.stream_ptr()),
                'hn_field_eval')

    field_launch()
    torch.cuda.synchronize()
    k_launches = max(1, min(args.steps, 3))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k_launches):
        field_launch()
    e1.record()
    torch.cuda.synchronize()
    kernel_ms_separate = e0.elapsed_time(e1) / k_launches
    # `roofline` is priced on the launches of the TIMED steps (above); the separate pass -- the same launch alone, after the fitting
    # legs have heated the part -- stays in the line as a cross-check
    kernel_ms = kernel_ms_in_loop if k_count.value == args.steps else kernel_ms_separate
    achieved = n * HAND_FLOP_PER_SAMPLE / (kernel_ms * 1e-3) / 1e12
    # ---- secondary figure: the same step with the exact far-field early-out enabled ("may additionally
    #      be reported culled"); `value` above stays the dense number
    culled = None
    if args.precision == 'f16x3' and not args.no_culled:
        field.set_culling(True)
        step()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        culled = samples_per_step * args.steps / (time.perf_counter() - tc)
        field.set_culling(False)
    # ---- secondary figure: the same step with the exact sample-level far-field skip (hn_field_set_compaction: the field runs on
    #      the compacted list of the samples that have a live bone mask; bit-identical output).  Never the headline.
    compact = None
    if args.precision == 'f16x3' and not args.no_culled:
        field.set_compaction(True)
        step()
        torch.cuda.synchronize()
        same = bool(torch.equal(step()['color_fine'], out['color_fine']))
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        compact = {'value': samples_per_step * args.steps / (time.perf_counter() - tc), 'bit_identical_to_dense': same}
        field.set_compaction(False)
    # ---- secondary figure (BASELINE configs[1] names "bf16"): the same step with precision = 'f16' -- hidden SDF layers and
    #      reverse sweep on ONE f16 MFMA per product (HN_PREC_F16) -- with its measured difference from the headline frame
    f16 = None
    if args.precision == 'f16x3' and not args.no_f16:
        ref_col, ref_ws = out['color_fine'].clone(), out['weight_sum'].clone()
        ren.precision = 'f16'
        o16 = step()
        torch.cuda.synchronize()
        t16 = time.perf_counter()
        for _ in range(args.steps):
            o16 = step()
        torch.cuda.synchronize()
        sec16 = (time.perf_counter() - t16) / args.steps
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        f16 = {'value': samples_per_step / sec16, 'unit': 'ray-samples/s', 'ms_per_step': sec16 * 1e3, 'dtype': 'f16 (one MFMA pass in the hidden SDF '
               'layers and the reverse sweep; encodings, feature layers, last SDF layer, colour network, alpha: fp32-equivalent)',
               'rel_diff_vs_f16x3': {'color_fine': rel(o16['color_fine'], ref_col), 'weight_sum': rel(o16['weight_sum'], ref_ws)},
               'error_vs_reference': 'pinned by tests/test_gpu_parity.py::test_f16_throughput_mode_error_is_pinned (sdf 3.4e-4, gradient 6.9e-4, rgb 3.6e-4)'}
        ren.precision = 'f16x3'
        out = step()
    if args.precision == 'f16x3':
        # three f16 MFMA products per fp32-equivalent product: the algorithmic rate is priced against a
        # third of the dense f16 MFMA peak (equivalently: issued MFMA FLOP/s against the full peak)
        peak, kname, dtype = PEAK_F16_MFMA_TFLOPS / 3.0, 'hn::v2::k_field2_hand<1>', 'f16x3'
    else:
        peak, kname, dtype = PEAK_F32_MFMA_TFLOPS, 'hn::k_field_hand<true>', 'f32'

    traffic, traffic_src, traffic_note = pmc_traffic(kname)
    if fitting is not None:
        # `roofline`: priced on the FLOP the step EXECUTES (the hand field runs on the samples with a live bone mask: the counts are
        # read from the compaction records of the timed configuration); `roofline_dense`: the same step with every sample evaluated
        # (single_12_dense), priced on the dense FLOP
        f16x3_peak = PEAK_F16_MFMA_TFLOPS / 3.0
        ex = fitting['single_12']['hand_samples_executed']
        sec = fitting['single_12']['ms_per_step'] * 1e-3
        flop = fit_step_flop(FIT_RAYS, hand_final=ex['final_evaluation'], hand_coarse=ex['coarse_pass'])
        fitting['roofline'] = {'bound': 'mfma', 'what': 'one fitting_single step (fit type 12), all kernels; FLOP of the samples the step executes '
                                                        '(exact far-field skip of the hand field on)', 'flop_per_step': flop,
                               'achieved': flop / sec / 1e12, 'peak': f16x3_peak, 'unit': 'TFLOP/s', 'frac': flop / sec / 1e12 / f16x3_peak}
        sec_d = fitting['single_12_dense']['ms_per_step'] * 1e-3
        flop_d = fit_step_flop(FIT_RAYS)
        fitting['roofline_dense'] = {'bound': 'mfma', 'what': 'the same step with every sample evaluated (single_12_dense)', 'flop_per_step': flop_d,
                                     'achieved': flop_d / sec_d / 1e12, 'peak': f16x3_peak, 'unit': 'TFLOP/s', 'frac': flop_d / sec_d / 1e12 / f16x3_peak}
        # the same priced on the C4 leg, where a rank fits its frames side by side (per frame-step: the whole leg / (frames x steps))
        sec_b = fitting['frames_sharded_12']['ms_per_step'] * 1e-3
        fitting['roofline_frames_side_by_side'] = {'bound': 'mfma', 'what': 'frames_sharded_12: per frame-step of the C4 leg (%d frames side by side), the FLOP of one '
                                                                              'executed fitting_single step' % fitting['frames_sharded_12'].get('frame_batch', 1),
                                                   'flop_per_step': flop, 'achieved': flop / sec_b / 1e12, 'peak': f16x3_peak, 'unit': 'TFLOP/s',
                                                   'frac': flop / sec_b / 1e12 / f16x3_peak}
        fitting['roofline']['traffic'], fitting['roofline']['traffic_note'] = fit_kernel_traffic()
        fitting['n_gpus'] = world
        if rccl_leg is not None:
            fitting['rccl_one_rank'] = rccl_leg
        fitting['config'] = ('C3/C4: fitting_single, %d rays x %d shared depths, both fields, 8 synthetic views, the reference\'s six-leaf pose chain, %d frames; '
                             'C5: fitting_video windows of %d frames x %d rays over a %d-frame sequence, fit type 1234, windows sharded over the GPUs with the '
                             'pose-gradient all-reduce' % (FIT_RAYS, FIT_N + 2 * FIT_IMP, C4_FRAMES, VID_FRAMES, VID_RAYS, C5_FRAMES))
        if not args.no_cpu_baseline:
            fitting['cpu_baseline'] = cpu_fit_baseline(single)
    training = None
    if not args.no_training and args.precision == 'f16x3':
        # one iteration of exp_runner.train (render, loss, backward into every network parameter, Adam,
        # re-pack) at the reference's batch (confs: 441 rays, 64 + 64 samples); secondary to the headline
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import train_step_bench
        training = {k: train_step_bench.measure(k, dev, 441, 10, 3, args.precision) for k in ('obj', 'hand')}
        # secondary: the hand iteration with every sample evaluated (the product's default aggregates the far field exactly)
        training['hand_dense'] = train_step_bench.measure('hand', dev, 441, 10, 3, args.precision, compact=False)
    c1 = time_c1(dev, args.precision, not args.no_cpu_baseline) if not args.no_c1 else None
    # ---- the rate the matrix pipe SUSTAINS on this box (hn_debug_mfma_probe: a kernel of nothing but f16 MFMAs on random
    #      operands, ~0.2 s per launch so that the power controller settles): the chip is power-limited well below the
    #      guide's 2.4 GHz figure, so the roofline entry states the fraction of this measured rate beside the nominal one
    sustained = None
    if args.precision == 'f16x3' and not share:
        import ctypes
        flop = ctypes.c_double(0.0)
        iters = 300000
        L.check(lib.hn_debug_mfma_probe(1, iters // 10, ctypes.byref(flop), L.stream_ptr()), 'hn_debug_mfma_probe')     # warm
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record()
        for _ in range(2):
            L.check(lib.hn_debug_mfma_probe(1, iters, ctypes.byref(flop), L.stream_ptr()), 'hn_debug_mfma_probe')
        p1.record()
        torch.cuda.synchronize()
        sustained = 2.0 * flop.value / (p0.elapsed_time(p1) * 1e-3) / 1e12        # TFLOP/s of issued f16 MFMA
    res = {
        'metric': 'ray-samples/sec/GPU (512x512x64)', 'value': value, 'unit': 'ray-samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
        'config': {'workload': 'C2: hand nets (conf size, random init), 512x512 rays x 64 samples, '
                               'n_importance=0, dense (no far-field culling), one frame per GPU',
                   'rays': B, 'samples_per_ray': N_SAMPLES, 'frames_per_step': world},
        'roofline': {'bound': 'mfma', 'kernel': kname, 'achieved': achieved,
                     'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                     'traffic': traffic, 'traffic_source': traffic_src, 'traffic_note': traffic_note, 'kernel_ms': kernel_ms,
                     'kernel_ms_what': ('mean over the %d field-evaluation launches of the timed steps themselves (HIP events on the launch stream, '
                                        'hn_debug_field_timer)' % k_count.value) if k_count.value == args.steps else 'separate pass (the in-loop timer saw %d launches)' % k_count.value,
                     'kernel_ms_separate_pass': kernel_ms_separate,
                     'flop_per_launch': n * HAND_FLOP_PER_SAMPLE,
                     'mfma_issued_tflops': achieved * (3.0 if args.precision == 'f16x3' else 1.0),
                     'mfma_peak_tflops': PEAK_F16_MFMA_TFLOPS if args.precision == 'f16x3' else PEAK_F32_MFMA_TFLOPS},
        'weight_sum_mean': float(out['weight_sum'].mean()),
    }
    if sustained is not None:
        # `peak` / `frac` above are the guide's nominal figures, as the contract asks; these two are measured in this run
        res['roofline']['mfma_sustained_tflops'] = sustained
        res['roofline']['frac_of_sustained'] = achieved * 3.0 / sustained
        res['roofline']['sustained_what'] = ('hn_debug_mfma_probe: v_mfma_f32_32x32x16_f16 alone on random operands, one wave per SIMD on every CU, '
                                              '2 x 0.2 s: the rate the power-limited matrix pipe holds on this box')
    if share:
        res['note'] = 'HONERF_BENCH_SHARE_GPU=1: all ranks on ONE device over gloo -- functional check only, timings are not a measurement'
    if c1 is not None:
        res['c1'] = c1
    if fitting is not None:
        res['fitting'] = fitting
    if training is not None:
        res['training'] = training
    if f16 is not None:
        res['value_f16'] = f16['value']      # secondary: never the headline (config C2 names bf16; parity needs fp32-equivalence)
        res['f16_mode'] = f16
    if culled is not None:
        res['value_culled'] = culled   # rank 0's frame, far-field early-out on (bit-identical output)
    if compact is not None:
        res['value_compact'] = compact['value']   # rank 0's frame, sample-level far-field skip on (bit-identical output)
        res['compact_bit_identical_to_dense'] = compact['bit_identical_to_dense']
    if not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(sdf, col, sc, args.cpu_crop)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

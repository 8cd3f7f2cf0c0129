#!/usr/bin/env python3
"""Times one iteration of exp_runner.train's inner loop on the native path (honerf_amd.training.train_step):
the reference's training batch (confs/wmask_realhand_hand1.conf: batch_size 441 rays = a 21 x 21 patch, 64 + 64 samples),
random-init networks of the conf shape, synthetic rays / targets.  Prints one JSON line per field kind with the step time
and its parts (render forward, backward incl. the parameter gradients, re-pack of the updated weights, optimiser)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(kind, dev):
    from honerf_amd import synth
    from honerf_amd.nets import (RenderingNetwork, RenderingNetwork_OBJ, SDFNetwork, SDFNetwork_OBJ, SingleVarianceNetwork)
    from honerf_amd.renderer import NeuSRenderer
    if kind == 'obj':
        sdf, col = SDFNetwork_OBJ().to(dev), RenderingNetwork_OBJ().to(dev)
    else:
        sdf, col = SDFNetwork().to(dev), RenderingNetwork(use_gradients=True).to(dev)
    var = SingleVarianceNetwork(0.3).to(dev)
    sdf.reset_parameters(21 if kind == 'hand' else 11)
    col.reset_parameters(22 if kind == 'hand' else 12)
    ren = NeuSRenderer(sdf, var, col, kind, 64, 64, 0, 4, 1.0)
    return ren, synth


def rays(kind, synth, n, dev, seed=3):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    rng = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    from honerf_amd import lib as L
    lib = L.load()
    if kind == 'hand':
        bt_inv, T_pose, joints = synth.synth_hand_pose(9)
        cam = synth.front_camera(dist=0.0, focal=2.0)
        tgt = joints[rng.randint(0, 21, size=n)] + 0.012 * rng.standard_normal((n, 3))
        xy = np.stack([tgt[:, 0] / tgt[:, 2] * 2.0, tgt[:, 1] / tgt[:, 2] * 2.0], -1).astype(np.float32)
        extra = dict(bt_inv=t(bt_inv), T_pose=t(T_pose), Ro=None, To=None)
    else:
        cam = synth.front_camera(dist=1.0, focal=2.0)
        xy = (rng.rand(n, 2).astype(np.float32) - 0.5) * 1.2
        R_obj, t_obj = synth.synth_obj_pose(2, center=(0.02, -0.01, 0.0))
        extra = dict(bt_inv=None, T_pose=None, Ro=t(R_obj.T.copy()), To=t(t_obj))
    o, d = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    c = {k: t(v) for k, v in cam.items()}
    L.check(lib.hn_ray_gen(L.ptr(t(xy)), L.ptr(c['R']), L.ptr(c['T']), L.ptr(c['focal']), L.ptr(c['principal']), 1, n, L.ptr(o), L.ptr(d),
                           L.stream_ptr()), 'hn_ray_gen')
    return o, d, extra


def measure(kind, dev, n_rays=441, steps=10, warmup=3, precision='f16x3', compact=True):
    """-> dict(workload, ms_per_step, ray_samples_per_s, parts_ms, loss, precision) for one field kind.  compact (hand only): the
    exact far-field aggregation of training.render_train (the product's default); False: every sample evaluated."""
    from honerf_amd import training
    ren, synth = build(kind, dev)
    ren.precision = precision
    ren.train_compact = bool(compact)
    ren.pack_eval_only = True
    o, d, ex = rays(kind, synth, n_rays, dev)
    g = torch.Generator(device='cpu').manual_seed(5)
    true_rgb = torch.rand(n_rays, 3, generator=g).to(dev)
    true_mask = (torch.rand(n_rays, 1, generator=g) > 0.3).float().to(dev)
    opt = training.make_optimizer(ren, 1e-4)          # exp_runner.py:107-110, confs learning_rate = 1e-4
    from honerf_amd import lib as _L
    _L.dropped_samples(reset=True)

    from honerf_amd.fitting import _unit_gradient as _unit

    def step(parts=None):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        ren.mark_parameters_changed()
        ren.field()                                   # re-pack of the weights the previous step updated
        ev[1].record()
        out = training.render_train(ren, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], None, ex['Ro'], ex['To'], repack=False)
        terms = training.train_loss(out, true_rgb, true_mask, 1.0, 1.0)
        ev[2].record()
        opt.zero_grad(set_to_none=True)
        terms['loss'].backward(gradient=_unit(terms['loss']))           # (as training.train_step)
        ev[3].record()
        opt.step()
        ev[4].record()
        if parts is not None:
            torch.cuda.synchronize()
            for i, k in enumerate(('repack', 'forward', 'backward', 'optimizer')):
                parts[k] = parts.get(k, 0.0) + ev[i].elapsed_time(ev[i + 1])
        return terms

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        terms = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    parts = {}
    for _ in range(steps):
        step(parts)
    S = 128
    # algorithmic work of the backward pass: four sweeps over the SDF network (tape, reverse, forward-direction, second
    # reverse), colour network forward + backward, and the outer products (two per SDF layer, one per colour layer)
    from honerf_amd import synth as _synth
    macs = lambda k: sum(o * i for o, i in _synth.layer_shapes(k, 256))
    p_sdf, p_col = macs('sdf_' + kind), macs('color_' + kind)
    fused = precision == 'f16x3' and os.environ.get('HN_TRAIN_FUSED', '1') != '0'
    kept_tape = fused and training.KEEP_TAPE
    # (with the tape kept by the forward pass the backward pass holds the adjoint's two sweeps and the colour network's backward only: the
    #  taped evaluation -- tape sweep, reverse sweep, colour forward -- is the render's own final evaluation)
    flop_sweeps = 2.0 * ((2 * p_sdf + p_col) if kept_tape else (4 * p_sdf + 2 * p_col)) * n_rays * S
    flop_outer = 2.0 * (2 * p_sdf + p_col) * n_rays * S
    flop = flop_sweeps + flop_outer
    bwd_s = parts['backward'] / steps * 1e-3
    if fused:
        # the pipes actually used: the sweeps in the fused f16x3 kernels (three fp16 MFMA passes per product: 2500 / 3 dense TFLOP/s of
        # fp32-equivalent work), the outer products in k_outer_group (six bf16 passes: 2500 / 6)
        peak = flop / (flop_sweeps / (2500.0 / 3) + flop_outer / (2500.0 / 6))
        what = ('backward pass: %sadjoint with the per-layer signals (k_field2_*<5>: f16x3, 3 MFMA passes) and the grouped outer products '
                '(k_outer_group: bf16, 6 passes); peak = the flop-weighted harmonic mean of 2500/3 and 2500/6 TFLOP/s' % (
                    '' if kept_tape else 'taped evaluation (k_field2_*<3>) + '))
    else:
        peak, what = 157.3, 'backward pass: launch sequence on v_mfma_f32_32x32x2_f32 (k_dense, k_outer)'
    roof = {'bound': 'mfma', 'what': what, 'flop_per_step': flop, 'achieved': flop / bwd_s / 1e12, 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': flop / bwd_s / 1e12 / peak}
    roof_hbm = None
    if fused:
        # ... and what the two kernels of that pass are actually bound by: HBM.  Algorithmic bytes of the backward pass per evaluated sample:
        # 41 KB of per-layer signals written by k_field2_*<5> + the tape it reads (the block the render kept) + ~58 KB the outer products
        # read (41 signal rows + X + GXb + feature vector, ~1.4 uses each; hand: X and GXb are 1 386 wide: + ~20 KB)
        from honerf_amd import lib as _L2
        tape_b = _L2.load().hn_render_single_tape_bytes(ren.field().handle, n_rays, S) if kept_tape else 0
        per_sample = 41 * 1024 + (58 if kind == 'obj' else 78) * 1024
        hbm_bytes = n_rays * S * per_sample + tape_b
        roof_hbm = {'bound': 'hbm', 'what': 'backward pass, algorithmic bytes: signals written + tape read + operands of the outer products read (dense sample count)',
                    'bytes_per_step': hbm_bytes, 'achieved': hbm_bytes / bwd_s / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': hbm_bytes / bwd_s / 1e9 / 8000.0}
        roof_hbm['traffic'], roof_hbm['traffic_note'] = train_kernel_traffic(kind if (compact or kind == 'obj') else kind + '_dense', kind)
    if compact and kind == 'hand':      # the work of the aggregated iteration depends on the batch's live fraction: priced in `hand_dense`
        roof = {'note': 'far-field aggregation on: fewer samples than the dense FLOP count assumes; the roofline entry is on the dense iteration (hand_dense)'}
        roof_hbm = None
    return {'roofline': roof, 'roofline_hbm': roof_hbm, 'workload': 'exp_runner.train iteration, %s nets, %d rays x (64+64) samples' % (kind, n_rays), 'ms_per_step': round(ms, 3),
            'far_field_aggregation': bool(compact and kind == 'hand'),
            'iterations_per_s': round(1e3 / ms, 2), 'ray_samples_per_s': round(n_rays * S / ms * 1e3),
            'parts_ms': {k: round(v / steps, 3) for k, v in parts.items()}, 'loss': float(terms['loss'].detach()), 'precision': precision,
            # samples the hand adjoint dropped (out of the fp16 fragments' range next to a bone's origin) over all %d iterations of this leg
            'dropped_samples': _L.dropped_samples(), 'dropped_samples_of': (warmup + 2 * steps) * n_rays * S}


def train_kernel_traffic(tag, kind):
    """HBM-side bytes per backward pass of its two kernels (k_field2_<kind><5> once, k_outer_group as often as the pass launches it), from the
    committed PMC summaries of tools/profile_train.sh (profiles/r*/pmc_train_<kind>_*.json: FETCH_SIZE / WRITE_SIZE in separate --pmc passes,
    FETCH_SIZE doubled for gfx950) -- quoted only if collected on the kernel sources of this tree.  -> (bytes or None, note)"""
    import glob
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    now = srchash.source_hash()
    dirs = sorted(d for d in glob.glob(os.path.join(ROOT, 'profiles', 'r*')) if os.path.exists(os.path.join(d, 'pmc_train_%s_k_outer_group.json' % tag)))
    if not dirs:
        return None, 'no PMC summary of the training kernels is committed'
    try:
        adj = json.load(open(os.path.join(dirs[-1], 'pmc_train_%s_k_field2_%s5.json' % (tag, kind))))
        out = json.load(open(os.path.join(dirs[-1], 'pmc_train_%s_k_outer_group.json' % tag)))
    except Exception:
        return None, 'incomplete PMC summaries in %s' % os.path.relpath(dirs[-1], ROOT)
    if adj.get('csrc_sha16') != now or out.get('csrc_sha16') != now:
        return None, 'STALE: %s was collected on kernel sources %s, this tree is %s' % (os.path.relpath(dirs[-1], ROOT), adj.get('csrc_sha16'), now)
    per_pass = out['dispatches']['FETCH_SIZE'] / float(adj['dispatches']['FETCH_SIZE'])     # k_outer_group launches per backward pass
    a_b, o_b = adj['derived']['hbm_bytes_per_launch'], out['derived']['hbm_bytes_per_launch'] * per_pass
    return a_b + o_b, ('k_field2_%s<5> %.3g + k_outer_group %.3g bytes per backward pass (%.3g launches of %.3g), '
                       '%s, csrc_sha16 %s' % (kind, a_b, o_b, per_pass, out['derived']['hbm_bytes_per_launch'], os.path.relpath(dirs[-1], ROOT), now))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=441)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--kinds', default='obj,hand')
    ap.add_argument('--out', default=None)
    ap.add_argument('--precision', default='f16x3')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lines = []
    for kind in a.kinds.split(','):
        line = measure(kind.split('_')[0], dev, a.rays, a.steps, a.warmup, a.precision, compact=not kind.endswith('_dense'))
        print(json.dumps(line))
        lines.append(line)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(lines, f, indent=1)


if __name__ == '__main__':
    main()

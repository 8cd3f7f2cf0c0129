"""HN_PREC_F16 (single-pass throughput mode) against the reference fixtures and the f16x3 kernels: errors + C2 kernel time.
   python tools/f16_mode_probe.py  (GPU box)"""
import os, sys, json, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import bench
from helpers import product_modules, rel_err, cu, t
from honerf_amd.nets import PackedField
from honerf_amd.renderer import NeuSRenderer
from honerf_amd import lib as L

res = {}
g = dict(np.load(os.path.join(R, 'tests', 'golden', 'field_hand.npz')))
m = product_modules()
for prec in ('f16x3', 'f16'):
    f = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision=prec)
    pts, dirs = cu(g['pts']), cu(g['dirs'])
    F_ = g['bt_inv'].shape[0] if g['bt_inv'].ndim == 4 else 1
    sdf, grad, rgb = f.evaluate(pts, dirs, 1, t(g['bt_inv']), t(g['T_pose']))
    ref = {'sdf': g['out'][:, :1], 'grad': g['grad'], 'rgb': g['rgb']}
    res['field_hand ' + prec] = {k: rel_err(v.cpu().numpy().reshape(ref[k].shape), ref[k]) for k, v in (('sdf', sdf), ('grad', grad), ('rgb', rgb))}
    sd = f.sdf(pts, t(g['bt_inv']), t(g['T_pose']))
    res['field_hand ' + prec]['sdf_only'] = rel_err(sd.cpu().numpy().reshape(-1, 1), ref['sdf'])
gr = dict(np.load(os.path.join(R, 'tests', 'golden', 'render_hand_64_0.npz')))
for prec in ('f16x3', 'f16'):
    ren = NeuSRenderer(m['sdf_hand'], m['var_hand'], m['color_hand'], 'hand', int(gr['n_samples']), 0, 0, 4, 1.0)
    ren.precision = prec
    out = ren.render(cu(gr['rays_o']), cu(gr['rays_d']), float(gr['near']), float(gr['far']), gr['bt_inv'], gr['T_pose'], None, None, None, 0,
                     t_rand=cu(gr['t_rand']))
    res['render_hand_64_0 ' + prec] = {k: rel_err(out[k].cpu().numpy().reshape(gr[k].shape), gr[k]) for k in ('color_fine', 'weight_sum', 'cdf_fine', 'weight_max')}
# C2 frame: both precisions, time + difference
dev = torch.device('cuda')
outs = {}
for prec in ('f16x3', 'f16'):
    ren, sdf, col, sc = bench.build_scene(dev, 9, prec)
    lib = L.load()
    B = bench.H_IMG * bench.W_IMG
    o, d = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    L.check(lib.hn_ray_gen(L.ptr(sc['xy']), L.ptr(sc['R']), L.ptr(sc['T']), L.ptr(sc['focal']), L.ptr(sc['principal']), 1, B, L.ptr(o), L.ptr(d), L.stream_ptr()), 'ray_gen')
    step = lambda: ren.render(o, d, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0, t_rand=sc['t_rand'])
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    outs[prec] = {k: out[k].clone() for k in ('color_fine', 'weight_sum')}
    res['C2 ' + prec] = {'ms_per_frame': dt * 1e3, 'ray_samples_per_s': B * 64 / dt}
res['C2 f16 vs f16x3'] = {k: rel_err(outs['f16'][k].cpu().numpy(), outs['f16x3'][k].cpu().numpy()) for k in outs['f16']}
print(json.dumps(res, indent=1))

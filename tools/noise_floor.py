"""How far is the fp32 reference itself from an fp64 evaluation of the same networks?
(establishes the conditioning-limited noise floor for the parity tolerances in DESIGN.md)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import oracle_fields, t, rel_err
from oracle import render as orr
hand, obj = oracle_fields()
def to64(f):
    f.sdf = [(W.double(), b.double()) for W, b in f.sdf]
    f.color = [(W.double(), b.double()) for W, b in f.color]
    return f
hand64, obj64 = to64(oracle_fields()[0]), to64(oracle_fields()[1])
for kind in ('obj', 'hand'):
    g = dict(np.load('tests/golden/render_%s_64_64.npz' % kind))
    o, d = t(g['rays_o']), t(g['rays_d'])
    if kind == 'obj':
        o, d = orr.obj_local(o, d, t(g['Ro']), t(g['To']))
    z = t(g['z_vals']); sd = (1.5 - 0.4) / 64
    mid, dists = orr.mid_points(z, sd)
    pts = orr._pts(o, d, mid).reshape(-1, 3)
    dirs = d[:, None, :].expand(z.shape[0], z.shape[1], 3).reshape(-1, 3)
    f32 = obj if kind == 'obj' else hand
    f64 = obj64 if kind == 'obj' else hand64
    kw = {} if kind == 'obj' else dict(bt_inv=t(g['bt_inv']), T_pose=t(g['T_pose']))
    kw64 = {k: v.double() for k, v in kw.items()}
    s32, g32, c32 = f32.evaluate(pts, dirs, **kw)
    s64, g64, c64 = f64.evaluate(pts.double(), dirs.double(), **kw64)
    for nm, a, b, gold in (('sdf', s32, s64, g['ps_sdf']), ('grad', g32, g64, g['ps_grad']), ('rgb', c32, c64, g['ps_rgb'])):
        a = a.detach().numpy(); b = b.detach().numpy()
        print('%s %-4s oracle32-vs-64 %.2e | reference32(golden)-vs-64 %.2e | oracle32-vs-golden %.2e | max|x| %.3g' % (
            kind, nm, rel_err(a, b), rel_err(gold.reshape(b.shape), b), rel_err(a, gold.reshape(a.shape)), np.abs(b).max()))

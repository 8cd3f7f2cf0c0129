import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd import lib as L
from oracle import render as orr
lib = L.load()
st = L.stream_ptr
G = lambda n: dict(np.load('tests/golden/%s.npz' % n))
# coarse z
B = 77
tr = torch.rand(B, 1)
for n in (64, 32, 40):
    z = torch.empty(B, n, device='cuda')
    L.check(lib.hn_coarse_z(L.ptr(cu(tr)), B, n, 0.4, 1.5, L.ptr(z), st()), 'cz')
    ref = orr.coarse_z(0.4, 1.5, n, tr).numpy()
    zz = z.cpu().numpy()
    d = zz != ref
    print('coarse n=%d mismatches %d cols %s maxulp %g' % (n, d.sum(), np.unique(np.where(d)[1])[:20], np.abs(zz.view(np.int32) - ref.view(np.int32)).max()))
    lin = torch.linspace(0, 1, n).numpy()
    step = np.float32(1) / np.float32(n - 1)
    k = np.arange(n)
    mine = np.where(k < n // 2, k.astype(np.float32) * step, np.float32(1) - (n - 1 - k).astype(np.float32) * step).astype(np.float32)
    print('  linspace formula mismatch idx', np.where(mine != lin)[0])
# upsample
g = G('upsample')
z, sdf = cu(g['z']), cu(g['sdf'])
Bu, k = z.shape
for i in range(1):
    z_new = torch.empty(Bu, 16, device='cuda')
    inds = torch.empty(Bu, 16, device='cuda', dtype=torch.int64)
    L.check(lib.hn_upsample(L.ptr(z), L.ptr(sdf), Bu, k, 16, 64.0, L.ptr(z_new), L.ptr(inds), st()), 'ups')
    a = z_new.cpu().numpy(); b = g['znew0']
    print('upsample ulp max', np.abs(a.view(np.int32) - b.view(np.int32)).max(), 'n diff', (a != b).sum(), 'of', a.size)
# render core on oracle depths
for kind in ('obj', 'hand'):
    g = G('render_%s_64_64' % kind)
    hand_o, obj_o = oracle_fields()
    field_o = obj_o if kind == 'obj' else hand_o
    kw = dict(Ro=t(g['Ro']), To=t(g['To'])) if kind == 'obj' else dict(bt_inv=t(g['bt_inv']), T_pose=t(g['T_pose']))
    ref = orr.render_single(field_o, t(g['rays_o']), t(g['rays_d']), 0.4, 1.5, t(g['t_rand']), 64, 64, 4, **kw)
    hand, obj = packed_fields()
    f = obj if kind == 'obj' else hand
    o, d = t(g['rays_o']), t(g['rays_d'])
    if kind == 'obj':
        o, d = orr.obj_local(o, d, t(g['Ro']), t(g['To']))
    B, S = ref['z_vals'].shape
    z = cu(ref['z_vals'])
    pts = torch.empty(B * S, 3, device='cuda'); dists = torch.empty(B * S, device='cuda')
    sd = (1.5 - 0.4) / 64
    L.check(lib.hn_sample_points(L.ptr(cu(o)), L.ptr(cu(d)), L.ptr(z), B, S, 1, sd, L.ptr(pts), L.ptr(dists), st()), 'pts')
    sdf, grad, rgb = f.evaluate(pts, cu(d), S, g.get('bt_inv'), g.get('T_pose'))
    for nm, a, b in (('sdf', sdf, ref['sdf']), ('grad', grad, ref['gradients']), ('rgb', rgb, ref['sampled_color'] if 'sampled_color' in ref else None)):
        if b is None: continue
        a = a.cpu().numpy().reshape(b.shape); b = b.detach().numpy()
        e = np.abs(a - b)
        print(kind, nm, 'max abs err %.3e at %s, max|ref| %.3e, ref there %s, got %s' % (e.max(), np.unravel_index(e.argmax(), e.shape), np.abs(b).max(), b.reshape(-1)[e.argmax()], a.reshape(-1)[e.argmax()]))
    print(kind, 'ref keys', list(ref.keys()))
    # float64 oracle?

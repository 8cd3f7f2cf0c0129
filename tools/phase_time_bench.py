"""Per-phase time of the hand field kernel on the points of the bench.py frame (C2), via the HN_DBG early exits.
   for k in 1 3 5 6 7 8 0; do HN_DBG=$((k*256)) python tools/phase_time_bench.py; done"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import lib as L
lib = L.load()
dev = torch.device('cuda')
ren, sdf, col, sc = bench.build_scene(dev, seed=9)
B, S = bench.H_IMG * bench.W_IMG, bench.N_SAMPLES
rays_o, rays_d = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
L.check(lib.hn_ray_gen(L.ptr(sc['xy']), L.ptr(sc['R']), L.ptr(sc['T']), L.ptr(sc['focal']), L.ptr(sc['principal']), 1, B,
                       L.ptr(rays_o), L.ptr(rays_d), L.stream_ptr()), 'ray_gen')
dbg = os.environ.pop('HN_DBG', '0')
out = ren.render(rays_o, rays_d, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0, t_rand=sc['t_rand'])
z = ren.last_z_vals
n = B * S
pts, dists = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
L.check(lib.hn_sample_points(L.ptr(rays_o), L.ptr(rays_d), L.ptr(z), B, S, 1, (bench.FAR - bench.NEAR) / S, L.ptr(pts), L.ptr(dists), L.stream_ptr()), 'sp')
field = ren.field()
o_sdf, o_grad, o_rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
wsb = lib.hn_field_workspace_bytes(field.handle, n)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
bt, tp = sc['bt_inv'].reshape(1, 21, 4, 4).contiguous(), sc['T_pose'].reshape(1, 21, 3).contiguous()
os.environ['HN_DBG'] = dbg
# one tile per workgroup: 256 x 128 consecutive samples (512 rays) from a chosen image row (default: the middle)
row = int(os.environ.get('ROW', '256'))
m = 32768
pts_s = pts[row * 512 * S: row * 512 * S + m].contiguous()
rd_s = rays_d[row * 512: row * 512 + m // S].contiguous()
def launch():
    L.check(lib.hn_field_eval(field.handle, L.ptr(pts_s), L.ptr(rd_s), m, S, L.ptr(bt), L.ptr(tp), 1, m, L.ptr(o_sdf), L.ptr(o_grad), L.ptr(o_rgb), None, L.ptr(ws), wsb, L.stream_ptr()), 'eval')
launch(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); launch(); launch(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
print('ROW=%d HN_DBG=%s: %.1f us (one 128-sample tile per workgroup)' % (row, dbg, ms * 1e3))

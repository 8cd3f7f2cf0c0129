import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, time
from helpers import *
from honerf_amd.nets import PackedField
from honerf_amd import lib as L
lib = L.load()
m = product_modules()
kind = sys.argv[1] if len(sys.argv) > 1 else 'obj'
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'
if kind == 'obj':
    f = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision=prec)
    fl_full, fl_sdf = 2 * (2 * 524544 + 292864), 2 * 524544
else:
    f = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision=prec)
    fl_full, fl_sdf = 2 * (2 * 1234176 + 624640), 2 * 1234176
gen = torch.Generator().manual_seed(3)
n = 1 << 21
from honerf_amd import synth
bt_inv, T_pose, joints = synth.synth_hand_pose(5)
if kind == 'obj':
    p = (torch.rand(n, 3, generator=gen) - 0.5) * 1.2
else:
    j = torch.from_numpy(joints)
    p = j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)
d = torch.nn.functional.normalize(torch.randn(n // 64, 3, generator=gen), dim=-1)
pc, dc = p.cuda(), d.cuda()
bt, tp = (torch.from_numpy(bt_inv).cuda().reshape(1, 21, 4, 4), torch.from_numpy(T_pose).cuda().reshape(1, 21, 3)) if kind == 'hand' else (None, None)
sdf, grad, rgb = torch.empty(n, device='cuda'), torch.empty(n, 3, device='cuda'), torch.empty(n, 3, device='cuda')
wsb = lib.hn_field_workspace_bytes(f.handle, n)
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
def full():
    L.check(lib.hn_field_eval(f.handle, L.ptr(pc), L.ptr(dc), n, 64, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(grad), L.ptr(rgb), None, L.ptr(ws), wsb, L.stream_ptr()), 'eval')
def sdfo():
    L.check(lib.hn_field_sdf(f.handle, L.ptr(pc), n, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(ws), wsb, L.stream_ptr()), 'sdf')
for name, fn, fl in (('full', full, fl_full), ('sdf', sdfo, fl_sdf)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 5e3
    print('%s %s %s HN_DBG=%s: %.2f ms  %.1f M samples/s  %.1f TFLOP/s (algorithmic)' % (kind, prec, name, os.environ.get('HN_DBG', '0'), dt * 1e3, n / dt / 1e6, n * fl / dt / 1e12))

"""One warm launch + a few timed launches of a field kernel, for rocprofv3 (--kernel-trace / --pmc)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import product_modules
from honerf_amd.nets import PackedField
from honerf_amd import lib as L, synth
lib = L.load()
kind = sys.argv[1] if len(sys.argv) > 1 else 'hand'
mode = sys.argv[2] if len(sys.argv) > 2 else 'full'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
m = product_modules()
f = PackedField(kind, m['sdf_' + kind], m['color_' + kind], m['var_' + kind], precision='f16x3')
gen = torch.Generator().manual_seed(3)
bt_inv, T_pose, joints = synth.synth_hand_pose(5)
if kind == 'obj':
    p = (torch.rand(n, 3, generator=gen) - 0.5) * 1.2
    bt = tp = None
else:
    j = torch.from_numpy(joints)
    p = j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)
    bt, tp = torch.from_numpy(bt_inv).cuda().reshape(1, 21, 4, 4), torch.from_numpy(T_pose).cuda().reshape(1, 21, 3)
d = torch.nn.functional.normalize(torch.randn(n // 64, 3, generator=gen), dim=-1)
pc, dc = p.cuda(), d.cuda()
sdf, grad, rgb = torch.empty(n, device='cuda'), torch.empty(n, 3, device='cuda'), torch.empty(n, 3, device='cuda')
wsb = lib.hn_field_workspace_bytes(f.handle, n)
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
for _ in range(3):
    if mode == 'full':
        L.check(lib.hn_field_eval(f.handle, L.ptr(pc), L.ptr(dc), n, 64, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(grad), L.ptr(rgb), None, L.ptr(ws), wsb, L.stream_ptr()), 'eval')
    else:
        L.check(lib.hn_field_sdf(f.handle, L.ptr(pc), n, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(ws), wsb, L.stream_ptr()), 'sdf')
torch.cuda.synchronize()
print('done', kind, mode, n)

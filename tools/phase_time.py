import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd.nets import PackedField
from honerf_amd import lib as L, synth
lib = L.load()
m = product_modules()
f = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='f16x3')
gen = torch.Generator().manual_seed(3)
n = 32768 * int(os.environ.get('TILES', '1'))
bt_inv, T_pose, joints = synth.synth_hand_pose(5)
j = torch.from_numpy(joints)
p = j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)
d = torch.nn.functional.normalize(torch.randn(n // 64, 3, generator=gen), dim=-1)
pc, dc = p.cuda(), d.cuda()
bt, tp = torch.from_numpy(bt_inv).cuda().reshape(1, 21, 4, 4), torch.from_numpy(T_pose).cuda().reshape(1, 21, 3)
sdf, grad, rgb = torch.empty(n, device='cuda'), torch.empty(n, 3, device='cuda'), torch.empty(n, 3, device='cuda')
wsb = lib.hn_field_workspace_bytes(f.handle, n)
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
def full():
    L.check(lib.hn_field_eval(f.handle, L.ptr(pc), L.ptr(dc), n, 64, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(grad), L.ptr(rgb), None, L.ptr(ws), wsb, L.stream_ptr()), 'eval')
full(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): full()
e1.record(); torch.cuda.synchronize()
print('HN_DBG=%s: %.3f ms per launch (1 tile per workgroup)' % (os.environ.get('HN_DBG', '0'), e0.elapsed_time(e1) / 5))

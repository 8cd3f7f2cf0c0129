import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd.nets import PackedField
m = product_modules()
kind = sys.argv[1] if len(sys.argv) > 1 else 'obj'
if kind == 'obj':
    f16 = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision='f16x3')
    f32 = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision='fp32')
    kw = {}
else:
    from honerf_amd import synth
    f16 = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='f16x3')
    f32 = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='fp32')
    bt_inv, T_pose, joints = synth.synth_hand_pose(5)
    kw = dict(bt_inv=bt_inv, T_pose=T_pose)
gen = torch.Generator().manual_seed(3)
for n in (32768, 32768 + 128, 40000, 65536, 70000):
    if kind == 'obj':
        p = (torch.rand(n, 3, generator=gen) - 0.5) * 1.2
    else:
        j = torch.from_numpy(joints)
        p = j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    a = f16.evaluate(cu(p), cu(d), 1, **kw); b = f32.evaluate(cu(p), cu(d), 1, **kw)
    for nm, x, y in zip(('sdf', 'grad', 'rgb'), a, b):
        x = x.cpu().numpy().reshape(n, -1); y = y.cpu().numpy().reshape(n, -1)
        e = np.abs(x - y).max(1)
        bad = np.where(e > 1e-5 * np.abs(y).max())[0]
        print(n, nm, 'rel err %.2e' % (e.max() / np.abs(y).max()), 'n_bad', len(bad), 'first bad', bad[:8], 'tiles', np.unique(bad // 128)[:10])
n = 32768
gen = torch.Generator().manual_seed(3)
p = (torch.rand(n, 3, generator=gen) - 0.5) * 1.2
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
a = f16.evaluate(cu(p), cu(d), 1, **kw); b = f32.evaluate(cu(p), cu(d), 1, **kw)
so = f16.sdf(cu(p), **kw).cpu().numpy().reshape(-1)
x = a[0].cpu().numpy().reshape(-1); y = b[0].cpu().numpy().reshape(-1)
bad = np.where(np.abs(x - y) > 1e-5 * np.abs(y).max())[0]
for i in bad[:14]:
    print(i, 'lane', i % 64, 'wave', (i // 32) % 4, 'p', p[i].numpy(), 'full16 %.7f sdfonly16 %.7f v1 %.7f' % (x[i], so[i], y[i]))

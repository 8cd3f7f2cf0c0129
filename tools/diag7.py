import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd.nets import PackedField
from honerf_amd import synth
m = product_modules()
f16 = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='f16x3')
f32 = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='fp32')
bt_inv, T_pose, joints = synth.synth_hand_pose(5)
gen = torch.Generator().manual_seed(3)
n = 32768
j = torch.from_numpy(joints)
p = j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
a = [x.cpu().numpy().reshape(n, -1) for x in f16.evaluate(cu(p), cu(d), 1, bt_inv, T_pose)]
b = [x.cpu().numpy().reshape(n, -1) for x in f32.evaluate(cu(p), cu(d), 1, bt_inv, T_pose)]
h64, _ = oracle_fields_fp64()
sel = np.argsort(-np.abs(a[2] - b[2]).max(1))[:64]
sel = np.concatenate([sel, np.arange(64)])
e = h64.evaluate(p[sel].double(), d[sel].double(), bt_inv=t(bt_inv).double(), T_pose=t(T_pose).double())
e = [x.detach().numpy().reshape(len(sel), -1) for x in e]
for k, nm in enumerate(('sdf', 'grad', 'rgb')):
    s = np.abs(e[k]).max()
    print(nm, 'worst-64 samples: f16x3-vs-fp64 %.2e  fp32kernel-vs-fp64 %.2e | first 64 samples: f16x3 %.2e fp32 %.2e   (rel to max %.3g)' % (
        np.abs(a[k][sel[:64]] - e[k][:64]).max() / s, np.abs(b[k][sel[:64]] - e[k][:64]).max() / s,
        np.abs(a[k][sel[64:]] - e[k][64:]).max() / s, np.abs(b[k][sel[64:]] - e[k][64:]).max() / s, s))
print('worst sample', sel[0], 'p', p[sel[0]].numpy(), 'grad fp64', e[1][0], 'f16x3', a[1][sel[0]], 'fp32', b[1][sel[0]])

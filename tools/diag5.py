import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd.nets import PackedField
m = product_modules()
f16 = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='f16x3')
f32 = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='fp32')
g = dict(np.load('tests/golden/field_hand.npz'))
pts, dirs = cu(g['pts']), cu(g['dirs'])
s = f16.sdf(pts, g['bt_inv'], g['T_pose'])
print('sdf-only rel err', rel_err(s.cpu().numpy(), g['out'][:, :1]))
sdf, grad, rgb, feat = f16.evaluate(pts, dirs, 1, g['bt_inv'], g['T_pose'], want_feat=True)
for nm, a, b in (('sdf', sdf, g['out'][:, :1]), ('feat', feat, g['out'][:, 1:]), ('grad', grad, g['grad']), ('rgb', rgb, g['rgb'])):
    print(nm, 'rel err vs golden', rel_err(a.cpu().numpy(), b))
b = f32.evaluate(pts, dirs, 1, g['bt_inv'], g['T_pose'])
for nm, x, y in zip(('sdf', 'grad', 'rgb'), (sdf, grad, rgb), b):
    print(nm, 'f16x3 vs fp32 kernel', rel_err(x.cpu().numpy(), y.cpu().numpy()))

import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd.nets import PackedField
from honerf_amd import lib as L
lib = L.load()
m = product_modules()
f = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='f16x3')
g = dict(np.load('tests/golden/field_hand.npz'))
n = 40
pts = cu(g['pts']); bt = cu(g['bt_inv']).reshape(1, 21, 4, 4); tp = cu(g['T_pose']).reshape(1, 21, 3)
sdf = torch.zeros(n, device='cuda')
wsb = lib.hn_field_workspace_bytes(f.handle, n)
ws = torch.full((wsb,), 255, dtype=torch.uint8, device='cuda')
L.check(lib.hn_field_sdf(f.handle, L.ptr(pts), n, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(ws), wsb, L.stream_ptr()), 'sdf')
torch.cuda.synchronize()
print('sdf', sdf.cpu().numpy()[:8], 'ref', g['out'][:8, 0])
w = ws.cpu().numpy()
SLOT = 32768
wave0 = w[:18 * SLOT]
feat = wave0[11 * SLOT:11 * SLOT + 87 * 2048].view(np.float16).reshape(87, 2, 64, 8).astype(np.float32)
print('FEAT nan count per block (hi):', np.isnan(feat[:, 0]).sum(axis=(1, 2))[:24], '...', np.isnan(feat[:, 0]).sum(), 'lo', np.isnan(feat[:, 1]).sum())
left = wave0[17 * SLOT:17 * SLOT + 21 * 256].view(np.float32).reshape(21, 64)
print('LEFT nan', np.isnan(left).sum(), left[:3, :4])
val = feat[:, 0] + feat[:, 1] / 2048
print('feature block 0 lane 0:', val[0, 0], ' lane 32:', val[0, 32])
print('max |feat|', np.nanmax(np.abs(val)))
a4f = wave0[10 * SLOT:11 * SLOT].view(np.float16).reshape(16, 2, 64, 8).astype(np.float32)
print('A4F nan per k-step', np.isnan(a4f).sum(axis=(1, 2, 3)))
print('LEFT region nan rows', np.isnan(left).sum(1))
allf = wave0[11 * SLOT:17 * SLOT].view(np.float16).reshape(96, 2, 64, 8).astype(np.float32)
print('FEAT region nan per block', np.isnan(allf).sum(axis=(1, 2, 3)))
big = np.argwhere(np.abs(feat) > 1.5)
print('n big', len(big), 'first', big[:10])
blk_bad = np.unique(big[:, 0]); print('blocks with big values', blk_bad)
print('hi/lo', np.unique(big[:, 1]), 'lanes', np.unique(big[:, 2])[:70], 'j', np.unique(big[:, 3]))
from oracle import nets as on
v_, r_, h_ = on.bone_coords(t(g['pts']), t(g['bt_inv']), t(g['T_pose']))
exp1 = (r_[:, :, 1] * h_[:, :, 0]).numpy()   # [n, 21]
exp2 = (r_[:, :, 2] * h_[:, :, 0]).numpy()
got = left   # [21, 64]
for b in (2, 10, 3):
    print('bone', b, 'lane 13 got', got[b, 13], 'exp', exp1[13, b], '| lane 45 got', got[b, 45], 'exp', exp2[13, b], '| lane 28 got', got[b, 28], 'exp', exp1[28, b])
print('max abs diff half0', np.abs(got[:, :32].T - exp1[:32]).max(), 'half1', np.abs(got[:, 32:].T - exp2[:32]).max())
raw = wave0[11 * SLOT:11 * SLOT + 87 * 2048].view(np.uint16).reshape(87, 2, 64, 8)
for lane_ in (13, 14):
    print('block 84 lane', lane_, 'hi', feat[84, 0, lane_], 'lo', feat[84, 1, lane_], 'raw hi', [hex(x) for x in raw[84, 0, lane_]])
    print('   expected values', exp1[lane_, :8])

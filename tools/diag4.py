import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, math
import torch.nn.functional as F
from helpers import *
from honerf_amd.nets import PackedField
from honerf_amd import lib as L
from oracle import nets as on
lib = L.load()
m = product_modules()
f16 = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision='f16x3')
f32 = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision='fp32')
gen = torch.Generator().manual_seed(3)
n0 = 32768
p = (torch.rand(n0, 3, generator=gen) - 0.5) * 1.2
d = torch.nn.functional.normalize(torch.randn(n0, 3, generator=gen), dim=-1)
a = f16.evaluate(cu(p), cu(d), 1)[0].cpu().numpy().reshape(-1)
b = f32.evaluate(cu(p), cu(d), 1)[0].cpu().numpy().reshape(-1)
bad = np.where(np.abs(a - b) > 1e-5 * np.abs(b).max())[0]
print('bad', bad)
sel = np.concatenate([bad[:16], np.arange(112)])[:128]
ps, ds = p[sel].contiguous(), d[sel].contiguous()
n = 128
pc, dc = ps.cuda(), ds.cuda()
sdf, grad, rgb = torch.empty(n, device='cuda'), torch.empty(n, 3, device='cuda'), torch.empty(n, 3, device='cuda')
wsb = lib.hn_field_workspace_bytes(f16.handle, n)
ws = torch.zeros(wsb, dtype=torch.uint8, device='cuda')
L.check(lib.hn_field_eval(f16.handle, L.ptr(pc), L.ptr(dc), n, 1, None, None, 1, n, L.ptr(sdf), L.ptr(grad), L.ptr(rgb), None, L.ptr(ws), wsb, L.stream_ptr()), 'eval')
torch.cuda.synchronize()
ref = f32.evaluate(pc, dc, 1)[0].cpu().numpy().reshape(-1)
print('sdf err in the 128-sample run:', np.abs(sdf.cpu().numpy() - ref)[:16])
w = ws.cpu().numpy().view(np.float32)
SLOTF = 8 * 4 * 64 * 4
NS = 11
def act(wave, slot):   # -> [256 neurons, 32 samples]
    base = (wave * NS + slot) * SLOTF
    x = w[base:base + SLOTF].reshape(8, 4, 64, 4)    # [t][q][lane][c]
    out = np.zeros((256, 32), np.float32)
    for t in range(8):
        for q in range(4):
            for c in range(4):
                i = 4 * q + c
                for hh in range(2):
                    row = (i & 3) + 8 * (i >> 2) + 4 * hh
                    out[32 * t + row] = x[t, q, 32 * hh:32 * hh + 32, c]
    return out
# oracle activations (fp64)
hand_o, obj_o = oracle_fields()
mlp = [(W.double(), bb.double()) for W, bb in obj_o.sdf]
inp = torch.cat([ps.double(), on.embed(ps.double(), 10)], -1)
x = inp
acts = []
for l, (W, bb) in enumerate(mlp[:8]):
    if l == 4:
        x = torch.cat([x, inp], 1) / math.sqrt(2.0)
    x = F.softplus(F.linear(x, W, bb), beta=100.0)
    acts.append(x.numpy())
for l in range(7):
    for wave in range(1):
        got = act(wave, l)            # a_{l+1}
        want = acts[l][wave * 32:wave * 32 + 32].T
        nn_ = want.shape[0]
        e = np.abs(got[:nn_] - want)
        print('layer a%d wave %d: max err per sample (first 16 = bad ones):' % (l + 1, wave), np.array2string(e.max(0)[:20], precision=2))
for (l, smp) in ((1, 11), (2, 6), (3, 0)):
    got = act(0, l); want = acts[l][0:32].T
    e = np.abs(got - want)[:, smp]
    idx = np.argsort(-e)[:8]
    print('layer a%d sample %d: worst neurons' % (l + 1, smp), idx, 'err', e[idx], 'want', want[idx, smp], 'got', got[idx, smp])
    print('   n(err>1e-6) =', (e > 1e-6).sum(), ' prev-layer input of this sample: min/max', acts[l - 1][smp].min(), acts[l - 1][smp].max())

import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, time
from helpers import *
from honerf_amd.nets import PackedField
m = product_modules()
obj = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision='f16x3')
obj32 = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'], precision='fp32')
g = dict(np.load('tests/golden/field_obj.npz'))
pts, dirs = cu(g['pts']), cu(g['dirs'])
s = obj.sdf(pts)
print('sdf-only rel err', rel_err(s.cpu().numpy(), g['out'][:, :1]))
sdf, grad, rgb, feat = obj.evaluate(pts, dirs, 1, want_feat=True)
for nm, a, b in (('sdf', sdf, g['out'][:, :1]), ('feat', feat, g['out'][:, 1:]), ('grad', grad, g['grad']), ('rgb', rgb, g['rgb'])):
    print(nm, 'rel err', rel_err(a.cpu().numpy(), b))
# ragged + big
gen = torch.Generator().manual_seed(3)
for n in (1, 31, 129, 1000, 70000):
    p = (torch.rand(n, 3, generator=gen) - 0.5) * 1.2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    a = obj.evaluate(cu(p), cu(d), 1); b = obj32.evaluate(cu(p), cu(d), 1)
    print(n, [rel_err(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(a, b)], 'sdf-only', rel_err(obj.sdf(cu(p)).cpu().numpy(), b[0].cpu().numpy()))
# timing
n = 1 << 21
p = (torch.rand(n, 3, generator=gen) - 0.5) * 1.2
d = torch.nn.functional.normalize(torch.randn(n // 64, 3, generator=gen), dim=-1)
pc, dc = cu(p), cu(d)
for f, name in ((obj, 'f16x3'), (obj32, 'fp32')):
    for full in (True, False):
        fn = (lambda: f.evaluate(pc, dc, 64)) if full else (lambda: f.sdf(pc))
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        fl = 2 * (2 * 524544 + 292864) if full else 2 * 524544
        print('%s %s: %.2f ms  %.1f M samples/s  %.1f TFLOP/s (algorithmic)' % (name, 'full' if full else 'sdf', dt * 1e3, n / dt / 1e6, n * fl / dt / 1e12))

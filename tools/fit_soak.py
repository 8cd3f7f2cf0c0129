"""Soak of the frame-sharded fitting loop: N frames x 200 steps (fit type 12, 8 views x 25 passes) twice with the same per-frame
seeds -- every leaf finite, the two runs equal to the bit (the losses printed are of different views: synthetic targets, random-init nets).  python tools/fit_soak.py [frames]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from honerf_amd import fitting as F
dev = torch.device('cuda')
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ren, nets, _, _, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True)
def run():
    out = {}
    t0 = time.perf_counter()
    for f in range(n_frames):
        ch, jf, _ = bench.build_fit_data(dev, 240 + f, 1, halo=True)
        views = F.synthetic_views(8, 1, bench.FIT_RAYS, 240 + f, jf[9], device=dev)
        torch.manual_seed(7000 + f)
        first = F.fit_step(ren, views[0], ch, F.make_optimizer(ch, video=False), bench.NEAR, bench.FAR, '12')   # (a throw-away optimiser: the loss at the start)
        l0 = float(first['loss'])
        ch, jf, _ = bench.build_fit_data(dev, 240 + f, 1, halo=True)
        torch.manual_seed(7000 + f)
        terms, steps = F.fit_frame(ren, views, ch, bench.NEAR, bench.FAR, '12')
        torch.cuda.synchronize()
        out[f] = ([p.detach().clone() for p in ch.parameters()], l0, float(terms['loss']), steps)
    return out, time.perf_counter() - t0
a, ta = run()
b, tb = run()
ok = True
for f in range(n_frames):
    fin = all(bool(torch.isfinite(p).all()) for p in a[f][0])
    same = all(torch.equal(x, y) for x, y in zip(a[f][0], b[f][0]))
    ok = ok and fin and same
    print('frame %d: %d steps, loss %.4f -> %.4f, finite %s, second run equal to the bit %s' % (f, a[f][3], a[f][1], a[f][2], fin, same))
print('%d frames x 2 runs: %.2f s / %.2f s; %s' % (n_frames, ta, tb, 'OK' if ok else 'FAILED'))
sys.exit(0 if ok else 1)

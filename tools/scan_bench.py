"""HBM roofline of the standalone stage kernels: algorithmic bytes per launch / measured time,
at the C2 image size (262144 rays; S = 64 or 128).  Prints one line per kernel and a JSON summary.
  python tools/scan_bench.py [out.json]"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from honerf_amd import lib as L
lib = L.load()
dev = torch.device('cuda')
B = 262144
PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md
g = torch.Generator(device='cuda').manual_seed(0)
def rnd(*s): return torch.rand(*s, device=dev, generator=g)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
rows = []
def report(name, bytes_, fn):
    dt = timeit(fn)
    gbs = bytes_ / dt / 1e9
    rows.append({'kernel': name, 'bytes_per_launch': bytes_, 'us': dt * 1e6, 'GB_per_s': gbs, 'frac_of_hbm_peak': gbs / PEAK})
    print('%-22s %8.1f us  %7.1f MB  %7.0f GB/s  %.2f of peak' % (name, dt * 1e6, bytes_ / 1e6, gbs, gbs / PEAK))
st = L.stream_ptr()
for S in (64, 128):
    n = B * S
    sdf, grad, dists = rnd(n) - 0.5, rnd(n, 3) - 0.5, rnd(n) * 0.02
    rays_d = torch.nn.functional.normalize(rnd(B, 3) - 0.5, dim=-1)
    alpha, c = torch.empty(n, device=dev), torch.empty(n, device=dev)
    report('hn_alpha S=%d' % S, n * (4 + 12 + 4 + 4 + 4) + B * 12,
           lambda: L.check(lib.hn_alpha(L.ptr(sdf), L.ptr(grad), L.ptr(rays_d), L.ptr(dists), n, S, 20.0, L.ptr(alpha), L.ptr(c), st), 'alpha'))
    rgb = rnd(n, 3)
    a2 = rnd(n) * 0.2
    color, w, ws_, wm, eik = torch.empty(B, 3, device=dev), torch.empty(n, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev), torch.zeros(2, device=dev)
    report('hn_composite1 S=%d' % S, n * (4 + 4 + 12 + 12 + 4) + B * 20,
           lambda: L.check(lib.hn_composite1(L.ptr(a2), L.ptr(c), L.ptr(rgb), L.ptr(grad), B, S, L.ptr(color), L.ptr(w), L.ptr(ws_), L.ptr(wm), L.ptr(eik), st), 'c1'))
    w2 = torch.empty(n, device=dev)
    rgb2, grad2 = rnd(n, 3), rnd(n, 3) - 0.5      # the second field's own arrays (re-using the first's would be served by L2)
    report('hn_composite2 S=%d' % S, n * 2 * (4 + 12 + 12 + 4) + B * 16,
           lambda: L.check(lib.hn_composite2(L.ptr(a2), L.ptr(rgb), L.ptr(grad), L.ptr(alpha), L.ptr(rgb2), L.ptr(grad2), B, S, L.ptr(color), L.ptr(ws_), L.ptr(w), L.ptr(w2), L.ptr(eik), st), 'c2'))
    ro = rnd(B, 3)
    z = torch.sort(rnd(B, S) * 1.1 + 0.4, dim=-1)[0].contiguous()
    pts, dd = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
    report('hn_sample_points S=%d' % S, B * 24 + n * (4 + 12 + 4),
           lambda: L.check(lib.hn_sample_points(L.ptr(ro), L.ptr(rays_d), L.ptr(z), B, S, 1, 1.1 / S, L.ptr(pts), L.ptr(dd), st), 'sp'))
for k in (64, 112):
    z = torch.sort(rnd(B, k) * 1.1 + 0.4, dim=-1)[0].contiguous()
    s = (z - 0.9).contiguous()
    zn, inds = torch.empty(B, 16, device=dev), torch.empty(B, 16, dtype=torch.int64, device=dev)
    report('hn_upsample k=%d' % k, B * (k * 8 + 16 * 4 + 16 * 8),
           lambda: L.check(lib.hn_upsample(L.ptr(z), L.ptr(s), B, k, 16, 64.0, L.ptr(zn), L.ptr(inds), st), 'up'))
    zn = torch.sort(rnd(B, 16) * 1.1 + 0.4, dim=-1)[0].contiguous()
    sn = (zn - 0.9).contiguous()
    zo, so, idx = torch.empty(B, k + 16, device=dev), torch.empty(B, k + 16, device=dev), torch.empty(B, k + 16, dtype=torch.int64, device=dev)
    report('hn_merge k=%d' % k, B * ((k + 16) * 8 * 2 + (k + 16) * 8),
           lambda: L.check(lib.hn_merge(L.ptr(z), L.ptr(zn), L.ptr(s), L.ptr(sn), B, k, 16, 0, L.ptr(zo), L.ptr(so), L.ptr(idx), st), 'merge'))
# the wave-per-ray form of up_sample (the fitting loops: <= 8192 rays), at the fitting size and at its largest batch
for Bw in (196, 8192):
    for k in (64, 112):
        z = torch.sort(rnd(Bw, k) * 1.1 + 0.4, dim=-1)[0].contiguous()
        s = (z - 0.9).contiguous()
        zn, inds = torch.empty(Bw, 16, device=dev), torch.empty(Bw, 16, dtype=torch.int64, device=dev)
        report('hn_upsample wave form, %d rays, k=%d' % (Bw, k), Bw * (k * 8 + 16 * 4 + 16 * 8),
               lambda: L.check(lib.hn_upsample(L.ptr(z), L.ptr(s), Bw, k, 16, 64.0, L.ptr(zn), L.ptr(inds), st), 'up'))
if len(sys.argv) > 1:
    json.dump({'rays': B, 'hbm_peak_GB_per_s': PEAK, 'kernels': rows}, open(sys.argv[1], 'w'), indent=1)

"""Per-chunk in-kernel stamps of the object field's taped evaluation (k_field2_obj<3>) and adjoint (<4>) launches; library built with
-DHN_TS (HONERF_LIB).  N_SAMPLES (default 13568 = 106 tiles, the CUs the object gets beside the hand in a fitting step)."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import packed_fields
from honerf_amd import lib as L
lib = L.load()
dev = torch.device('cuda')
n = int(os.environ.get('N_SAMPLES', '13568'))
spr = 64
gen = torch.Generator().manual_seed(0)
pts = ((torch.rand(n, 3, generator=gen) - 0.5) * 0.9).to(dev).contiguous()
d = torch.nn.functional.normalize(torch.randn(n // spr, 3, generator=gen), dim=-1).to(dev).contiguous()
gs, gg, gr = (torch.randn(n, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev))
_, obj = packed_fields('cuda', 'f16x3')
sdf, grad, rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
tb = lib.hn_field_tape_bytes(obj.handle, n)
tape = torch.empty(tb, dtype=torch.uint8, device=dev)
wsb = lib.hn_field_workspace_bytes(obj.handle, n)
ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
bwb = lib.hn_field_bwd_workspace_bytes(obj.handle, n)
bws = torch.empty(bwb, dtype=torch.uint8, device=dev)
g_pts, g_d = torch.empty(n, 3, device=dev), torch.empty(n // spr, 3, device=dev)
raw = ctypes.CDLL(L.LIB_PATH)
raw.hn_debug_ts_obj.argtypes = [ctypes.c_void_p, ctypes.c_int]
def taped():
    L.check(lib.hn_field_eval_taped(obj.handle, L.ptr(pts), L.ptr(d), n, spr, None, None, 1, n, L.ptr(sdf), L.ptr(grad), L.ptr(rgb), L.ptr(ws), wsb, L.ptr(tape), tb,
                                    L.stream_ptr()), 'taped')
def adj():
    L.check(lib.hn_field_eval_bwd_taped(obj.handle, L.ptr(pts), L.ptr(d), n, spr, None, None, 1, n, L.ptr(gs), L.ptr(gg), L.ptr(gr), L.ptr(grad), L.ptr(rgb),
                                        L.ptr(tape), L.ptr(g_pts), L.ptr(g_d), None, None, L.ptr(bws), bwb, L.stream_ptr()), 'adj')
def report(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (ctypes.c_ulonglong * (4 * 8192))()
    assert raw.hn_debug_ts_obj(buf, 4 * 8192) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 8192)
    ids = (a[0] >> np.uint64(60)).astype(int); t = (a[0] & np.uint64((1 << 60) - 1)).astype(np.int64)
    k = int(np.argmax(ids == 0)) if (ids == 0).any() else 8192
    ids, t = ids[:k], t[:k]
    print('%s: launch %.3f ms; wave 0: %d stamps spanning %d ticks -> %.1f ticks/us' % (name, ms, k, t[-1] - t[0], (t[-1] - t[0]) / (ms * 1e3)))
    starts = np.nonzero(ids == 1)[0]
    rows = []
    for ci, s in enumerate(starts):
        e = starts[ci + 1] if ci + 1 < len(starts) else k
        tt, ii = t[s:e], ids[s:e]
        t1 = tt[0]; t2 = tt[ii == 2][0]; t3 = tt[ii == 3][0]
        t4 = tt[ii == 4]
        last = t4[-1] if len(t4) else t3
        nxt = t[e] if e < k else last
        rows.append((t2 - t1, t3 - t2, last - t3, nxt - last))
    r = np.array(rows)
    print('  chunks', len(r), 'totals: dma-wait %d barrier %d mma %d tail %d' % tuple(r.sum(0)))
    for i in range(0, len(r), 25):
        s = r[i:i + 25].sum(0)
        print('  chunks %4d..%4d: dma %7d bar %7d mma %8d tail %8d | per chunk %5d' % (i, min(i + 25, len(r)) - 1, *s, r[i:i + 25].sum() / len(r[i:i + 25])))
    print('  largest tails', [(int(i), int(r[i, 3])) for i in np.argsort(-r[:, 3])[:10]])
    print('  largest dma waits', [(int(i), int(r[i, 0])) for i in np.argsort(-r[:, 0])[:10]])
report('k_field2_obj<3>', taped)
report('k_field2_obj<4>', adj)

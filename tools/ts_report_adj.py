"""Per-chunk in-kernel stamps of the hand field's evaluation + adjoint launch (k_field2_hand<2>, hn_field_eval_bwd) on a small
launch; library built with -DHN_TS (HONERF_LIB).  N_SAMPLES (default 3200 = 25 tiles)."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import packed_fields
from honerf_amd import lib as L, synth
lib = L.load()
dev = torch.device('cuda')
n = int(os.environ.get('N_SAMPLES', '3200'))
spr = 64
gen = torch.Generator().manual_seed(0)
bt, tp, j = synth.synth_hand_pose(5)
bt, tp = torch.from_numpy(bt)[None].to(dev).contiguous(), torch.from_numpy(tp)[None].to(dev).contiguous()
jt = torch.from_numpy(j)
pts = (jt[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)).to(dev).contiguous()
d = torch.nn.functional.normalize(torch.randn(n // spr, 3, generator=gen), dim=-1).to(dev).contiguous()
gs, gg, gr = (torch.randn(n, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev))
hand, obj = packed_fields('cuda', 'f16x3')
g_pts, g_d = torch.empty(n, 3, device=dev), torch.empty(n // spr, 3, device=dev)
g_bt, g_tp = torch.zeros(1, 21, 4, 4, device=dev), torch.zeros(1, 21, 3, device=dev)
need = lib.hn_field_bwd_workspace_bytes(hand.handle, n)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
def bwd():
    L.check(lib.hn_field_eval_bwd(hand.handle, L.ptr(pts), L.ptr(d), n, spr, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(gs), L.ptr(gg), L.ptr(gr),
                                  L.ptr(g_pts), L.ptr(g_d), L.ptr(g_bt), L.ptr(g_tp), L.ptr(ws), need, L.stream_ptr()), 'bwd')
for _ in range(3): bwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bwd(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * (4 * 8192))()
raw.hn_debug_ts_adj.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.hn_debug_ts_adj(buf, 4 * 8192) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 8192)
ids = (a[0] >> np.uint64(60)).astype(int); t = (a[0] & np.uint64((1 << 60) - 1)).astype(np.int64)
k = int(np.argmax(ids == 0)) if (ids == 0).any() else 8192
ids, t = ids[:k], t[:k]
print('launch %.3f ms; wave 0: %d stamps spanning %d ticks -> %.1f ticks/us' % (ms, k, t[-1] - t[0], (t[-1] - t[0]) / (ms * 1e3)))
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
print('chunks', len(r), 'totals: dma-wait %d barrier %d mma %d tail %d' % tuple(r.sum(0)))
for i in range(0, len(r), 50):
    s = r[i:i + 50].sum(0)
    print('chunks %4d..%4d: dma %7d bar %7d mma %8d tail %8d | per chunk %5d' % (i, min(i + 50, len(r)) - 1, *s, r[i:i + 50].sum() / len(r[i:i + 50])))
print('largest tails', [(int(i), int(r[i, 3])) for i in np.argsort(-r[:, 3])[:16]])
print('largest dma waits', [(int(i), int(r[i, 0])) for i in np.argsort(-r[:, 0])[:12]])

# sections of the adjoint (stamps 10 .. 15 at their starts; the tile's end is the last stamp)
names = {10: 'seeds + colour backward (lin4..1^T, lin0^T feature-vector / enc(g) rows)', 11: 'pass A: colour lin0^T over the feature rows, bone by bone',
         12: 'J gb as fragments', 13: 'forward-direction sweep', 14: 'second reverse sweep', 15: 'pass B: input map, pose gradients'}
sec = [(int(ids[x]), int(t[x])) for x in range(k) if ids[x] >= 10]
if sec:
    first = sec[0][1]
    print('evaluation part of the tile: %d ticks' % (first - int(t[0])))
    for i, (sid, ts) in enumerate(sec):
        te = sec[i + 1][1] if i + 1 < len(sec) else int(t[-1])
        nch = int(((ids == 1) & (t >= ts) & (t < te)).sum())
        print('  %-78s %8d ticks  %4d chunks  %6d per chunk' % (names.get(sid, str(sid)), te - ts, nch, (te - ts) // max(nch, 1)))

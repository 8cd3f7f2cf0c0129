"""Timing aid: per-chunk stamps of workgroup 0 of the hand field kernel (library built with -DHN_TS).
   make -C ho-nerf_amd/csrc clean; make -C ho-nerf_amd/csrc CXXFLAGS_EXTRA=-DHN_TS; python tools/ts_report.py"""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from helpers import *
from honerf_amd.nets import PackedField
from honerf_amd import lib as L, synth
lib = L.load()
m = product_modules()
f = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'], precision='f16x3')
gen = torch.Generator().manual_seed(3)
n = int(os.environ.get('N_SAMPLES', str(32768 * int(os.environ.get('TILES', '1')))))   # N_SAMPLES=3200: the 25-tile launches of a fitting step
bt_inv, T_pose, joints = synth.synth_hand_pose(5)
j = torch.from_numpy(joints)
p = j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)
d = torch.nn.functional.normalize(torch.randn(max(n // 64, 1), 3, generator=gen), dim=-1)
pc, dc = p.cuda(), d.cuda()
bt, tp = torch.from_numpy(bt_inv).cuda().reshape(1, 21, 4, 4), torch.from_numpy(T_pose).cuda().reshape(1, 21, 3)
sdf, grad, rgb = torch.empty(n, device='cuda'), torch.empty(n, 3, device='cuda'), torch.empty(n, 3, device='cuda')
wsb = lib.hn_field_workspace_bytes(f.handle, n)
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
def full():
    L.check(lib.hn_field_eval(f.handle, L.ptr(pc), L.ptr(dc), n, 64, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf), L.ptr(grad), L.ptr(rgb), None, L.ptr(ws), wsb, L.stream_ptr()), 'eval')
for _ in range(3): full()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); full(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * (4 * 8192))()
raw.hn_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.hn_debug_ts(buf, 4 * 8192) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 8192)
out = []
for w in range(4):
    ids = (a[w] >> np.uint64(60)).astype(int); t = (a[w] & np.uint64((1 << 60) - 1)).astype(np.int64)
    k = int(np.argmax(ids == 0)) if (ids == 0).any() else 8192
    ids, t = ids[:k], t[:k]
    span = t[-1] - t[0]
    if w == 0: print('launch %.3f ms; wave 0: %d stamps spanning %d ticks -> %.1f ticks/us' % (ms, k, span, span / (ms * 1e3)))
    # chunks: id 1 starts one
    starts = np.nonzero(ids == 1)[0]
    rows = []
    for ci, s in enumerate(starts):
        e = starts[ci + 1] if ci + 1 < len(starts) else k
        tt, ii = t[s:e], ids[s:e]
        t1 = tt[0]; t2 = tt[ii == 2][0]; t3 = tt[ii == 3][0]
        t4 = tt[ii == 4]
        last = t4[-1] if len(t4) else t3
        nxt = t[e] if e < k else last
        rows.append((t2 - t1, t3 - t2, last - t3, nxt - last, len(t4)))
    out.append(np.array(rows))
np.save(os.path.join(R, 'gpurun_out', 'ts_rows.npy'), np.array(out, dtype=object), allow_pickle=True)
r = out[0]
print('chunks:', len(r))
print('wave0 per chunk [dma-wait, barrier, mma, tail, tiles]:')
for ci in range(len(r)):
    print(ci, ' '.join('%5d' % x for x in r[ci]), '|', ' '.join('%5d' % sum(out[w][ci][:4]) for w in range(4) if ci < len(out[w])))

if os.environ.get('RAW'):
    w = 0
    ids = (a[w] >> np.uint64(60)).astype(int); t = (a[w] & np.uint64((1 << 60) - 1)).astype(np.int64)
    lo, hi = [int(x) for x in os.environ['RAW'].split(':')]
    starts = np.nonzero(ids == 1)[0]
    for ci in range(lo, hi):
        s0, e0 = starts[ci], starts[ci + 1]
        print('chunk', ci, ' '.join('%d:%d' % (ids[k], t[k] - t[s0]) for k in range(s0, e0)))

# tile prologue (stamps 6..9 of the timing build): tile start -> points known -> bone loop done -> features done -> first acquire
if os.environ.get('PROLOGUE'):
    ids0 = (a[0] >> np.uint64(60)).astype(int); t0 = (a[0] & np.uint64((1 << 60) - 1)).astype(np.int64)
    k0 = int(np.argmax(ids0 == 0)) if (ids0 == 0).any() else 8192
    ids0, t0 = ids0[:k0], t0[:k0]
    for s6 in np.nonzero(ids0 == 6)[0]:
        seg = [(int(ids0[x]), int(t0[x])) for x in range(s6, min(s6 + 6, k0))]
        prev4 = [x for x in range(max(s6 - 3, 0), s6) if ids0[x] == 4]
        base = t0[prev4[-1]] if prev4 else t0[s6]
        print('tile start: since last mma end %6d | ' % (t0[s6] - base) + ' '.join('id%d +%d' % (i, tt - t0[s6]) for i, tt in seg[1:]))

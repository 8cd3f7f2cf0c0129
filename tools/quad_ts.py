"""In-kernel phase stamps of the latency-form sdf kernel (a -DHN_QTS build of libhonerf: HONERF_LIB=...): cycles per phase of
workgroup 0.  python tools/quad_ts.py [n_blocks]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import packed_fields, t
from honerf_amd import lib as L, synth
lib = L.load()
hand, _ = packed_fields('cuda', 'f16x3')
bt_inv, T_pose, joints = synth.synth_hand_pose(7)
bt, tp = t(bt_inv)[None].cuda(), t(T_pose)[None].cuda()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 98
n = 32 * nb
gen = torch.Generator().manual_seed(3)
pts = (t(joints)[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)).cuda().contiguous()
L.check(lib.hn_debug_quad_max_blocks(1 << 20), 'q')
for _ in range(3):
    hand.sdf(pts, bt, tp)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
raw = ctypes.CDLL(L.LIB_PATH)
raw.hn_debug_qts(buf, 64)
names = ['F0 generation', 'F0 stores -> barrier -> leftover frags', 'lin0 feature pass', 'publish 0 (2 epilogues + exchange)', 'lin1 MFMAs', 'publish 1',
         'lin2 + lin3 (+ 2 publishes)', 'lin4 hidden part', 'lin4 feature pass', 'publish', 'lin5 .. lin7 (+ 2 publishes)', 'last epilogue + sdf']
ts = list(buf)
tot = ts[12] - ts[0]
for i, nm in enumerate(names):
    d = ts[i + 1] - ts[i]
    print('%-45s %8d cycles  %5.1f %%' % (nm, d, 100.0 * d / tot))
print('block total %d cycles' % tot)

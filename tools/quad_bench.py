"""Latency-form vs throughput sdf-only hand kernel at small launch sizes: us per launch.  python tools/quad_bench.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import packed_fields, t, cu
from honerf_amd import lib as L, synth
lib = L.load()
hand, obj = packed_fields('cuda', 'f16x3')
bt_inv, T_pose, joints = synth.synth_hand_pose(7)
bt, tp = t(bt_inv)[None].cuda(), t(T_pose)[None].cuda()
gen = torch.Generator().manual_seed(3)
for n_blocks in (1, 8, 32, 98, 196, 256, 392, 512):
    n = 32 * n_blocks
    j = t(joints)
    pts = (j[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)).cuda().contiguous()
    out = torch.empty(n, device='cuda')
    wsb = lib.hn_field_workspace_bytes(hand.handle, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    res = []
    for mb in (0, 1 << 20):
        L.check(lib.hn_debug_quad_max_blocks(mb), 'q')
        call = lambda: L.check(lib.hn_field_sdf(hand.handle, L.ptr(pts), n, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(out), L.ptr(ws), wsb, L.stream_ptr()), 'sdf')
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    L.check(lib.hn_debug_quad_max_blocks(-1), 'q')
    print('blocks %4d (%5d samples): throughput form %7.1f us   latency form %7.1f us' % (n_blocks, n, res[0], res[1]))

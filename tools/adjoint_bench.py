"""Time hn_field_eval / hn_field_eval_bwd of both fields at the fitting size (196 rays x 192 depths = 37 632 samples).
   python tools/adjoint_bench.py [out.json]"""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from helpers import packed_fields
from honerf_amd import lib as L, synth

lib = L.load()
dev = torch.device('cuda')
rays, spr = 196, 192
n = rays * spr
gen = torch.Generator().manual_seed(0)
bt, tp, j = synth.synth_hand_pose(5)
bt, tp = torch.from_numpy(bt)[None].to(dev).contiguous(), torch.from_numpy(tp)[None].to(dev).contiguous()
jt = torch.from_numpy(j)
pts_h = (jt[torch.randint(0, 21, (n,), generator=gen)] + 0.03 * torch.randn(n, 3, generator=gen)).to(dev).contiguous()
pts_o = ((torch.rand(n, 3, generator=gen) - 0.5) * 0.9).to(dev).contiguous()
d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=gen), dim=-1).to(dev).contiguous()
gs, gg, gr = (torch.randn(n, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev))
res = {}
for prec in ('f16x3', 'fp32'):
    hand, obj = packed_fields('cuda', prec)
    for name, f, pts in (('obj', obj, pts_o), ('hand', hand, pts_h)):
        isb = name == 'hand'
        g_pts, g_d = torch.empty(n, 3, device=dev), torch.empty(rays, 3, device=dev)
        g_bt, g_tp = torch.zeros(1, 21, 4, 4, device=dev), torch.zeros(1, 21, 3, device=dev)
        need = lib.hn_field_bwd_workspace_bytes(f.handle, n)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        sdf, grad, rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        needf = lib.hn_field_workspace_bytes(f.handle, n)
        wsf = torch.empty(max(needf, 16), dtype=torch.uint8, device=dev)

        def bwd():
            L.check(lib.hn_field_eval_bwd(f.handle, L.ptr(pts), L.ptr(d), n, spr, L.ptr(bt) if isb else None, L.ptr(tp) if isb else None, 1, n,
                                          L.ptr(gs), L.ptr(gg), L.ptr(gr), L.ptr(g_pts), L.ptr(g_d), L.ptr(g_bt) if isb else None,
                                          L.ptr(g_tp) if isb else None, L.ptr(ws), need, L.stream_ptr()), 'bwd')

        def fwd():
            L.check(lib.hn_field_eval(f.handle, L.ptr(pts), L.ptr(d), n, spr, L.ptr(bt) if isb else None, L.ptr(tp) if isb else None, 1, n,
                                      L.ptr(sdf), L.ptr(grad), L.ptr(rgb), None, L.ptr(wsf), needf, L.stream_ptr()), 'fwd')

        for tag, fn in (('fwd', fwd), ('bwd', bwd)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res['%s_%s_%s_ms' % (prec, name, tag)] = e0.elapsed_time(e1) / 10
        res['%s_%s_bwd_workspace_MB' % (prec, name)] = need / 1e6
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)

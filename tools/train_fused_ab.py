"""hn_field_param_bwd of an f16x3 OBJECT field through the fused path (k_field2_obj<3> + <5> + the outer products; default) against the
generic launch sequence (HN_TRAIN_FUSED=0), in two child processes: every parameter-gradient block, g_pts, g_rays_d, and the time.
   python tools/train_fused_ab.py [n_points]"""
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))


def child(n, out, kind='obj'):
    import numpy as np
    import torch
    from helpers import product_modules
    from honerf_amd import lib as L
    from honerf_amd.nets import PackedField
    lib = L.load()
    dev = torch.device('cuda')
    m = product_modules()
    pf = PackedField(kind, m['sdf_' + kind], m['color_' + kind], m['var_' + kind], precision='f16x3')
    g = torch.Generator().manual_seed(5)
    S = 64
    bt = tp = g_bt = g_tp = None
    if kind == 'obj':
        pts = ((torch.rand(n, 3, generator=g) - 0.5) * 0.9).to(dev)
    else:
        from honerf_amd import synth
        bt_np, tp_np, joints = synth.synth_hand_pose(9)
        off = 0.012 * torch.randn(n, 3, generator=g)
        off = off * (off.norm(dim=1, keepdim=True).clamp(min=float(os.environ.get('AB_MIN_DIST', '0.006'))) / off.norm(dim=1, keepdim=True))
        pts = torch.from_numpy(joints).float()[torch.randint(0, 21, (n,), generator=g)] + off
        pts[::8] += 0.5
        pts = pts.to(dev)
        bt, tp = torch.from_numpy(bt_np).float().reshape(1, 21, 4, 4).to(dev), torch.from_numpy(tp_np).float().reshape(1, 21, 3).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n // S, 3, generator=g), dim=-1).to(dev)
    gs, gg, gr = torch.randn(n, generator=g).to(dev), (torch.randn(n, 3, generator=g) * 0.1).to(dev), torch.randn(n, 3, generator=g).to(dev)
    need = lib.hn_field_bwd_workspace_bytes(pf.handle, n)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    nf = lib.hn_field_param_floats(pf.handle)

    def run():
        g_params = torch.zeros(nf, device=dev)
        g_pts, g_dir = torch.empty(n, 3, device=dev), torch.zeros(n // S, 3, device=dev)
        g_bt, g_tp = torch.zeros(1, 21, 4, 4, device=dev), torch.zeros(1, 21, 3, device=dev)
        L.check(lib.hn_field_param_bwd(pf.handle, L.ptr(pts), L.ptr(dirs), n, S, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(gs), L.ptr(gg), L.ptr(gr), L.ptr(g_params),
                                       L.ptr(g_pts), L.ptr(g_dir), L.ptr(g_bt), L.ptr(g_tp), L.ptr(ws), need, L.stream_ptr()), 'hn_field_param_bwd')
        return g_params, g_pts, g_dir, g_bt
    out_t = run()
    torch.cuda.synchronize()
    # the adjoint without parameter gradients (k_field2_obj<full>: what the fitting steps run)
    g_pts2, g_dir2 = torch.empty(n, 3, device=dev), torch.zeros(n // S, 3, device=dev)
    g_bt2, g_tp2 = torch.zeros(1, 21, 4, 4, device=dev), torch.zeros(1, 21, 3, device=dev)
    L.check(lib.hn_field_eval_bwd(pf.handle, L.ptr(pts), L.ptr(dirs), n, S, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(gs), L.ptr(gg), L.ptr(gr), L.ptr(g_pts2), L.ptr(g_dir2),
                                  L.ptr(g_bt2), L.ptr(g_tp2), L.ptr(ws), need, L.stream_ptr()), 'hn_field_eval_bwd')
    torch.cuda.synchronize()
    # ... and the taped pair (k_field2_*<3>, <4>)
    tb = lib.hn_field_tape_bytes(pf.handle, n)
    tape = torch.empty(tb, dtype=torch.uint8, device=dev)
    sdf_t, grad_t, rgb_t = torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    g_pts3, g_dir3 = torch.empty(n, 3, device=dev), torch.zeros(n // S, 3, device=dev)
    g_bt3, g_tp3 = torch.zeros(1, 21, 4, 4, device=dev), torch.zeros(1, 21, 3, device=dev)
    L.check(lib.hn_field_eval_taped(pf.handle, L.ptr(pts), L.ptr(dirs), n, S, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(sdf_t), L.ptr(grad_t), L.ptr(rgb_t), L.ptr(ws), need,
                                    L.ptr(tape), tb, L.stream_ptr()), 'hn_field_eval_taped')
    L.check(lib.hn_field_eval_bwd_taped(pf.handle, L.ptr(pts), L.ptr(dirs), n, S, L.ptr(bt), L.ptr(tp), 1, n, L.ptr(gs), L.ptr(gg), L.ptr(gr), L.ptr(grad_t), L.ptr(rgb_t),
                                        L.ptr(tape), L.ptr(g_pts3), L.ptr(g_dir3), L.ptr(g_bt3), L.ptr(g_tp3), L.ptr(ws), need, L.stream_ptr()), 'hn_field_eval_bwd_taped')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    np.savez(out, g_params=out_t[0].cpu().numpy(), g_pts=out_t[1].cpu().numpy(), g_dir=out_t[2].cpu().numpy(), ms=ms, ws=need, g_pts3=g_pts3.cpu().numpy(), g_bt3=g_bt3.cpu().numpy(), g_bt=out_t[3].cpu().numpy(), g_bt2=g_bt2.cpu().numpy(), pts=pts.cpu().numpy(), g_pts2=g_pts2.cpu().numpy(), g_dir2=g_dir2.cpu().numpy())


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 'obj')
        sys.exit(0)
    import numpy as np
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 56448
    kind = sys.argv[2] if len(sys.argv) > 2 else 'obj'
    outs = {}
    for flag in ('1', '0'):
        path = '/tmp/train_fused_ab_%s.npz' % flag
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', str(n), path, kind], env=dict(os.environ, HN_TRAIN_FUSED=flag))
        outs[flag] = np.load(path)
    a, b = outs['1'], outs['0']
    rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30))
    print('n = %d: fused %.2f ms (workspace %.2f GB), generic %.2f ms (%.2f GB)' % (n, a['ms'], a['ws'] / 1e9, b['ms'], b['ws'] / 1e9))
    print('g_params rel diff %.3e (max |g| %.3e), g_pts %.3e, g_rays_d %.3e' % (rel(a['g_params'], b['g_params']), np.abs(b['g_params']).max(),
                                                                             rel(a['g_pts'], b['g_pts']), rel(a['g_dir'], b['g_dir'])))
    print('against the adjoint without parameter gradients: fused g_pts %.3e, generic g_pts %.3e; fused g_dir %.3e, generic g_dir %.3e' % (
        rel(a['g_pts'], a['g_pts2']), rel(b['g_pts'], b['g_pts2']), rel(a['g_dir'], a['g_dir2']), rel(b['g_dir'], b['g_dir2'])))
    print('finite: fused g_params %s, generic g_params %s; fused g_pts %s; pose gradients fused vs adjoint %.3e, generic vs adjoint %.3e' % (
        np.isfinite(a['g_params']).all(), np.isfinite(b['g_params']).all(), np.isfinite(a['g_pts']).all(), rel(a['g_bt'], a['g_bt2']), rel(b['g_bt'], b['g_bt2'])))
    dg = np.abs(b['g_pts'] - a['g_pts2']).max(axis=1)
    worst = np.argsort(-dg)[:6]
    print('generic vs adjoint, worst samples %s: |diff| %s, |g| %s, zero in adjoint %s' % (worst, dg[worst], np.abs(b['g_pts'][worst]).max(axis=1), (a['g_pts2'][worst] == 0).all(axis=1)))
    print('against the TAPED pair: fused g_pts %.3e (%d samples differ at all), pose gradients %.3e; taped pair vs the one-launch adjoint g_pts %.3e' % (
        rel(a['g_pts'], a['g_pts3']), int((a['g_pts'] != a['g_pts3']).any(axis=1).sum()), rel(a['g_bt'], a['g_bt3']), rel(a['g_pts3'], a['g_pts2'])))
    d = np.abs(a['g_pts'] - a['g_pts2']).max(axis=1)
    bad = np.nonzero(d > 1e-3 * np.abs(a['g_pts2']).max())[0]
    print('fused vs adjoint: %d samples off, first %s, tiles %s' % (bad.size, bad[:16], sorted(set((bad // 128).tolist()))[:16]))
    # per 64 K-float block of the parameter vector: where a difference sits
    gp_a, gp_b = a['g_params'], b['g_params']
    worst = []
    for i in range(0, gp_a.size, 65536):
        x, y = gp_a[i:i + 65536], gp_b[i:i + 65536]
        worst.append((float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30)), i))
    worst.sort(reverse=True)
    print('worst blocks (rel diff, offset):', ['%.2e @ %d' % w for w in worst[:6]])

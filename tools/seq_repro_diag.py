"""Where do two runs of the SAME fitting_video sequence part?  (round 5: the one-rank RCCL leg found the sequence loop's result to
differ between runs of one process by ~2e-3 in the leaves, while a window's steps are bit-reproducible in isolation.)

Runs fit_sequence_video K times from identical initial state and seeds, snapshotting the six leaves ON THE DEVICE before every step
(clones on the caller's stream: no host synchronisation, the step's timing is not disturbed), then reports for every pair of runs the
first step whose snapshot differs, which leaves differ there and by how much.  --sync: the same with a device synchronisation after
every step (if that removes the differences, they come from launches of neighbouring steps overlapping)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from honerf_amd import fitting as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--runs', type=int, default=3)
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--sync', action='store_true')
    ap.add_argument('--no-side', action='store_true', help='fitting.USE_SIDE_STREAM = False')
    ap.add_argument('--no-compact', action='store_true')
    ap.add_argument('--no-defer', action='store_true', help='fitting.DEFER_JACOBIAN = False: the Jacobian launch IN FRONT of the stable term on the extra stream (as until round 5)')
    ap.add_argument('--check', action='store_true', help='hn_stable_pts against a torch restatement issued right behind it on the same stream')
    ap.add_argument('--no-aux-jac', action='store_true', help='pose.JACOBIAN_ON_AUX = False')
    ap.add_argument('--no-stable', action='store_true', help="fit type '123': no stable term")
    args = ap.parse_args()
    dev = torch.device('cuda')
    if args.no_defer:
        F.DEFER_JACOBIAN = False
    if args.check:
        from honerf_amd import autograd as AG0
        AG0._DIAG_CHECK[0] = True
    if args.no_side:
        F.USE_SIDE_STREAM = False
    if args.no_aux_jac:
        from honerf_amd import pose
        pose.JACOBIAN_ON_AUX = False
    renb, _ = bench.build_fit_nets(dev, bench.VID_FRAMES, 'f16x3')
    if args.no_compact:
        renb.compact_far_field = False
    runs = []
    fwd_log = []
    orig_step_loss = F.step_loss

    def spy(render_out, true_rgb, true_mask, pose, fit_type='1', video=False, smooth_ends=(False, False), stable=None, pose_terms=None):
        rec = {k: render_out[k].detach().clone() for k in ('color_fine', 'weight_sum', 'sdf_hand', 'sdf_obj')}
        rec.update({k: pose[k].detach().clone() for k in ('bt_inv', 'joint_3d', 'obj_r', 'obj_t')})
        if stable is not None:
            rec['stable'] = (stable if isinstance(stable, torch.Tensor) else stable.value).detach().clone().reshape(1)
            if not isinstance(stable, torch.Tensor):
                for nm, x in zip(('st_p', 'st_pw', 'st_bt', 'st_tp', 'st_dsdf', 'st_grad', 'st_rgb'), stable.saved):
                    rec[nm] = x.detach().clone()
        from honerf_amd import autograd as AG
        if AG._DIAG:
            d = AG._DIAG.pop()
            rec['chk_pw_vs_ref_on_side'] = d['pw_vs_ref_on_side'].clone()
            rec['chk_R_side_minus_R_main'] = (d['R_side'].reshape(-1) - d['obj_r'].detach().reshape(-1)).abs().max().reshape(1)
            rec['chk_t_side_minus_t_main'] = (d['t_side'].reshape(-1) - d['obj_t'].detach().reshape(-1)).abs().max().reshape(1)
        fwd_log[-1].append(rec)
        return orig_step_loss(render_out, true_rgb, true_mask, pose, fit_type, video, smooth_ends, stable, pose_terms)
    F.step_loss = spy
    for r in range(args.runs):
        fwd_log.append([])
        torch.manual_seed(77)
        chain, j, v = bench.build_fit_data(dev, 60, args.frames, halo=True, drift=0.002)
        ov = v[None].expand(bench.VID_FRAMES, -1, -1).contiguous()
        per_window = {tuple(w): F.synthetic_views(args.views, bench.VID_FRAMES, bench.VID_RAYS, 300 + w[0], j[9], device=dev)
                      for w in F.sliding_windows(args.frames)}
        snaps, meta, grads = [], [], []

        def window_views(index, vid, step):
            if args.sync:
                torch.cuda.synchronize()
            snaps.append(torch.cat([p.detach().reshape(-1) for p in chain.parameters()]).clone())
            grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1) for p in chain.parameters()]).clone())
            meta.append((tuple(index), vid, step))
            return per_window[tuple(index)][vid]
        window_views.n_views = args.views
        F.fit_sequence_video(renb, window_views, chain, bench.NEAR, bench.FAR, args.frames, '123' if args.no_stable else '1234', outer_iters=1, obj_verts=ov)
        torch.cuda.synchronize()
        snaps.append(torch.cat([p.detach().reshape(-1) for p in chain.parameters()]).clone())
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1) for p in chain.parameters()]).clone())
        runs.append((torch.stack(snaps).cpu(), meta, torch.stack(grads).cpu()))
    for a in range(args.runs):
        for b in range(a + 1, args.runs):
            for k, (ra, rb) in enumerate(zip(fwd_log[a], fwd_log[b])):
                bad = ['%s %.2e' % (key, float((ra[key] - rb[key]).abs().max())) for key in ra if not torch.equal(ra[key], rb[key])]
                if bad:
                    print('   run %d checks at that step: %s' % (a, {kk: float(vv) for kk, vv in ra.items() if kk.startswith('chk_')}))
                    print('   run %d checks at that step: %s' % (b, {kk: float(vv) for kk, vv in rb.items() if kk.startswith('chk_')}))
                    print('runs %d and %d: first differing FORWARD quantities at step %d: %s' % (a, b, k, '; '.join(bad)))
                    break
    n = args.frames
    names = ['obj_rot', 'obj_trans', 'palm_rot', 'palm_trans', 'joint_refine_angle', 'palm_refine_angle']
    sizes = [6 * n, 3 * n, 6 * n, 3 * n, 20 * n, 7 * n]
    for a in range(args.runs):
        for b in range(a + 1, args.runs):
            sa, sb = runs[a][0], runs[b][0]
            diff = (sa != sb).any(dim=1)
            if not bool(diff.any()):
                print('runs %d and %d: identical in all %d snapshots' % (a, b, sa.shape[0]))
                continue
            k = int(torch.nonzero(diff)[0])
            ga, gb = runs[a][2], runs[b][2]
            gd = (ga != gb).any(dim=1)
            kg = int(torch.nonzero(gd)[0]) if bool(gd.any()) else -1
            goff, gwhere = 0, []
            for nm, sz in zip(names, sizes):
                if kg >= 0:
                    d = (ga[kg, goff:goff + sz] - gb[kg, goff:goff + sz]).abs()
                    if float(d.max()) > 0:
                        gwhere.append('%s %.2e rel %.1e' % (nm, float(d.max()), float(d.max() / ga[kg, goff:goff + sz].abs().max().clamp_min(1e-30))))
                goff += sz
            print('   first differing GRADIENT: the one held before step %d (produced by step %d): %s' % (kg, kg - 1, '; '.join(gwhere)))
            off, where = 0, []
            for nm, sz in zip(names, sizes):
                d = (sa[k, off:off + sz] - sb[k, off:off + sz]).abs()
                if float(d.max()) > 0:
                    where.append('%s %.2e (%d entries)' % (nm, float(d.max()), int((d > 0).sum())))
                off += sz
            print('runs %d and %d: first difference BEFORE step %d of %d (i.e. produced by step %d: window %s view %d); %s; final max diff %.2e'
                  % (a, b, k, sa.shape[0] - 1, k - 1, runs[a][1][k - 1][0] if k > 0 else None, runs[a][1][k - 1][1] if k > 0 else -1, '; '.join(where),
                     float((sa[-1] - sb[-1]).abs().max())))


if __name__ == '__main__':
    main()

"""fitting_video window step with / without the exact far-field skip, and the live fraction of its samples.
python tools/video_compaction_probe.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F, lib as L
dev = torch.device('cuda')
cut = torch.tensor([0.08, 0.03, 0.03, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02], device=dev)
for compact in (True, False, True, False):
    renb, netsb, chainb, viewsb, ov = bench.build_fit(dev, 60, bench.VID_FRAMES, bench.VID_RAYS, 'f16x3', halo=True)
    renb.compact_far_field = compact
    optb = F.make_optimizer(chainb, video=True)
    idx = list(range(bench.VID_FRAMES))
    def step(i):
        F.fit_step(renb, viewsb[i % 8], chainb, optb, bench.NEAR, bench.FAR, '1234', index=idx, smooth_ends=(True, False), obj_verts_for_stable=ov)
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40): step(i)
    torch.cuda.synchronize()
    print('compaction %s: ms per window step %.3f' % (compact, (time.perf_counter() - t0) / 40 * 1e3))
v = viewsb[0]
pose = chainb(idx)
P = bench.VID_RAYS
o, d = F._rays(L, v['xy'], v['cam'], 4, P)
with torch.no_grad():
    out = renb.render(o.reshape(4, P, 3), d.reshape(4, P, 3), bench.NEAR, bench.FAR, pose['bt_inv'].detach(), pose['T_pose_21'], None,
                      torch.inverse(pose['obj_r']).detach(), pose['obj_t'].detach())
    z = renb.last_z_vals.reshape(4, P, -1)
    dist = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], (bench.FAR - bench.NEAR) / 64)], -1)
    pts = o.reshape(4, P, 1, 3) + d.reshape(4, P, 1, 3) * (z + 0.5 * dist)[..., None]
    bt = pose['bt_inv'].detach()
    q = torch.einsum('fbij,fnsj->fnsbi', bt[:, :, :3, :3], pts) + bt[:, None, None, :, :3, 3] - pose['T_pose_21'][:, None, None]
    hh = 1.0 - 1.0 / (1.0 + torch.exp(-200.0 * (q.norm(dim=-1) - cut)))
    live = (hh != 0).any(-1)
    print('live samples %.1f %% (%d of %d = %d tiles of 128)' % (100 * live.float().mean(), int(live.sum()), live.numel(), (int(live.sum()) + 127) // 128))

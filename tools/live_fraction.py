"""Fraction of the samples of a fitting step's final evaluation that have at least one live bone (mask h != 0):
what an exact far-field skip (SURVEY B-11) could remove.  python tools/live_fraction.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F, lib as L
dev = torch.device('cuda')
ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True)
cut = torch.tensor([0.08, 0.03, 0.03, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02], device=dev)
opt = F.make_optimizer(chain, video=False)
for step in range(8):
    v = views[step % 8]
    pose = chain()
    o, d = F._rays(L, v['xy'], v['cam'], 1, bench.FIT_RAYS)
    with torch.no_grad():
        out = ren.render(o, d, bench.NEAR, bench.FAR, pose['bt_inv'][0].detach(), pose['T_pose_21'][0], None, pose['obj_r'][0].T.detach(), pose['obj_t'][0].detach())
        z = ren.last_z_vals                                                   # [196, 192]
        dist = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], (bench.FAR - bench.NEAR) / 64)], -1)
        pts = o[:, None, :] + d[:, None, :] * (z + 0.5 * dist)[..., None]     # mid points
        bt = pose['bt_inv'][0].detach()
        q = torch.einsum('bij,nsj->nsbi', bt[:, :3, :3], pts) + bt[:, :3, 3] - pose['T_pose_21'][0]
        vv = q.norm(dim=-1)
        hh = 1.0 - 1.0 / (1.0 + torch.exp(-200.0 * (vv - cut)))
        live = (hh != 0).any(-1)
        tiles = (live.reshape(-1).float().reshape(-1, 128).sum(1) > 0).sum()
        print('view %d: live samples %.1f %% (%d of %d = %d tiles of 128 if compacted); tiles with any live sample now: %d of %d'
              % (step, 100 * live.float().mean(), int(live.sum()), live.numel(), (int(live.sum()) + 127) // 128, int(tiles), live.numel() // 128))
    F.fit_step(ren, v, chain, opt, bench.NEAR, bench.FAR, '12')

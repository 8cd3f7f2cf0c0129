"""How many of the 21 bones are live (mask h != 0 for at least one sample) in a 32-sample block of the compacted list of a fitting
step -- what a per-block skip of dead bones' weight blocks could remove from the latency-form sdf kernel.  python tools/live_bones_per_block.py"""
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
        z = ren.last_z_vals
        pts = o[:, None, :] + d[:, None, :] * z[..., None]
        bt = pose['bt_inv'][0].detach()
        q = torch.einsum('bij,nsj->nsbi', bt[:, :3, :3], pts) + bt[:, :3, 3] - pose['T_pose_21'][0]
        hh = 1.0 - 1.0 / (1.0 + torch.exp(-200.0 * (q.norm(dim=-1) - cut)))
        lb = (hh != 0).reshape(-1, 21)
        live = lb.any(-1)
        for name, sel in (('final 192 depths, compacted', lb[live]), ('every 12th depth (a 16-depth round), compacted', lb.reshape(196, 192, 21)[:, ::12].reshape(-1, 21)[live.reshape(196, 192)[:, ::12].reshape(-1)])):
            n = sel.shape[0] // 32 * 32
            blocks = sel[:n].reshape(-1, 32, 21).any(1).sum(-1).float()
            print('view %d  %-48s %5d live samples, %4d blocks: live bones per block mean %.1f  min %d  max %d' % (step, name, sel.shape[0], blocks.numel(), blocks.mean(), blocks.min(), blocks.max()))
    F.fit_step(ren, v, chain, opt, bench.NEAR, bench.FAR, '12')

"""Time of the pose side of a fitting step alone (pose chain forward, joint / vertex losses, backward, Adam): no render.
A stand-in gradient arrives at bt_inv / obj_r / obj_t as the renderer's would."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F
dev = torch.device('cuda')
for halo in (False, True):
    ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=halo)
    opt = F.make_optimizer(chain, video=False)
    gb, gr, gt = torch.randn(1, 21, 4, 4, device=dev) * 1e-3, torch.randn(1, 3, 3, device=dev) * 1e-3, torch.randn(1, 3, device=dev) * 1e-3
    def step():
        pose = chain()
        loss = (pose['bt_inv'] * gb).sum() + (pose['obj_r'] * gr).sum() + (pose['obj_t'] * gt).sum()
        from honerf_amd.pose import VertsLossFn
        jl = pose['joint_loss'][0] if 'joint_loss' in pose else F.pose_loss(pose['joint3d_pred'][0], pose['joint_3d'][0])
        loss = loss + 30.0 * jl + 20.0 * VertsLossFn.apply(pose['obj_r'], pose['obj_t'], pose['Ro_pred'], pose['To_pred'], pose['obj_verts'])[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize()
    print('%s chain: pose side of a step %.3f ms' % ('halo' if halo else 'rigid', (time.perf_counter() - t0) / 50 * 1e3))

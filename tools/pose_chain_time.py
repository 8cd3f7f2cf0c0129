import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from honerf_amd.pose import PoseChainFn
g = np.load('/root/repo/tests/golden/pose_chain.npz')
dev = torch.device('cuda')
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
for F in (1, 4, 32):
    ori, bl = t(np.repeat(g['ori_pose'][1:2], F, 0)), t(np.repeat(g['bone_len'][1:2], F, 0))
    prm = t(np.repeat(g['params'][1:2], F, 0)).requires_grad_(True)
    for _ in range(3):
        bt, j3 = PoseChainFn.apply(ori, bl, prm); (bt.sum() + j3.sum()).backward()
    torch.cuda.synchronize()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e0.record()
    for _ in range(20): bt, j3 = PoseChainFn.apply(ori, bl, prm)
    e1.record()
    for _ in range(20):
        bt, j3 = PoseChainFn.apply(ori, bl, prm); (bt.sum() + j3.sum()).backward()
    e2.record(); torch.cuda.synchronize()
    print('F=%d: forward (values + Jacobian) %.1f us, forward + backward %.1f us' % (F, e0.elapsed_time(e1) * 50, e1.elapsed_time(e2) * 50))

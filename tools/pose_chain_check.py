import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from honerf_amd import lib as L
lib = L.load()
g = np.load('/root/repo/tests/golden/pose_chain.npz')
dev = 'cuda'
f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
ori, bl, prm = f32(g['ori_pose']), f32(g['bone_len']), f32(g['params'])
F = ori.shape[0]
outs = []
for fill in (float('nan'), 7.0):
    for with_jac in (False, True):
        bt = torch.full((F, 21, 4, 4), fill, device=dev); j3 = torch.full((F, 21, 3), fill, device=dev)
        jac = torch.full((F, 399, 36), fill, device=dev) if with_jac else None
        L.check(lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, L.ptr(bt), L.ptr(j3), L.ptr(jac) if with_jac else None, L.stream_ptr()), 'pc')
        torch.cuda.synchronize()
        print('fill', fill, 'jac', with_jac, 'nan in bt', int(torch.isnan(bt).sum()), 'j3', int(torch.isnan(j3).sum()), 'jac', int(torch.isnan(jac).sum()) if with_jac else '-',
              'untouched(7.0) bt', int((bt == 7.0).sum()), 'j3', int((j3 == 7.0).sum()))
        outs.append((bt.clone(), j3.clone()))
print('bt equal across runs:', all(torch.equal(outs[0][0], o[0]) for o in outs[1:]) if not torch.isnan(outs[0][0]).any() else 'nan')
# rigid pose
out = torch.full((F, 412), float('nan'), device=dev)
jac = torch.full((F, 412, 18), float('nan'), device=dev)
po = torch.zeros(F, 18, device=dev); po[:, 0] = 1; po[:, 3] = 1
Ro = torch.eye(3, device=dev)[None].repeat(F, 1, 1).contiguous(); To = torch.zeros(F, 3, device=dev)
L.check(lib.hn_rigid_pose(None, None, L.ptr(Ro), L.ptr(To), L.ptr(po), F, 0, L.ptr(out), L.ptr(jac), L.stream_ptr()), 'rp')
torch.cuda.synchronize()
print('rigid (no palm): nan in out[399:411]', int(torch.isnan(out[:, 399:411]).sum()), 'nan in jac rows 399:411', int(torch.isnan(jac[:, 399:411]).sum()))

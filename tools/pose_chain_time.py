import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np, time
from honerf_amd import lib as L
lib=L.load(); dev=torch.device('cuda')
g=np.load('tests/golden/pose_chain.npz')
t=lambda a: torch.tensor(a,dtype=torch.float32,device=dev).contiguous()
ori,bl,prm=t(g['ori_pose'])[:1],t(g['bone_len'])[:1],t(g['params'])[:1]
bt,j3,jac=torch.empty(1,21,4,4,device=dev),torch.empty(1,21,3,device=dev),torch.empty(1,399,36,device=dev)
st=L.stream_ptr()
for name,args in (('joint',(L.ptr(bt),L.ptr(j3),L.ptr(jac))),('values',(L.ptr(bt),L.ptr(j3),None)),('jac',(None,None,L.ptr(jac)))):
    for _ in range(5): lib.hn_pose_chain(L.ptr(ori),L.ptr(bl),None,L.ptr(prm),1,*args,st)
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): lib.hn_pose_chain(L.ptr(ori),L.ptr(bl),None,L.ptr(prm),1,*args,st)
    b.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(b)/50*1000,'us')

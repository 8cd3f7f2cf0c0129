"""320 iterations of honerf_amd.training.train_step on a fixed hand batch: the loss goes down and the device memory in use
stays constant (the re-pack recycles its blocks; measured 5 368 MB at iterations 20, 120, 220, 320)."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import train_step_bench as tb
from honerf_amd import training
dev = torch.device('cuda:0')
for kind in ('hand',):
    ren, synth = tb.build(kind, dev)
    o, d, ex = tb.rays(kind, synth, 441, dev)
    g = torch.Generator().manual_seed(5)
    rgb, mask = torch.rand(441, 3, generator=g).to(dev), (torch.rand(441, 1, generator=g) > 0.3).float().to(dev)
    opt = training.make_optimizer(ren, 1e-4)
    tr = torch.rand(441, 1, generator=g).to(dev)
    for it in range(321):
        t = training.train_step(ren, opt, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], ex['Ro'], ex['To'], rgb, mask, 1.0, 1.0, t_rand=tr)
        if it in (20, 120, 220, 320):
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            print('%s iter %d loss %.4f  device memory in use %.1f MB  torch reserved %.1f MB' % (kind, it, float(t['loss'].detach()), (total - free) / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)

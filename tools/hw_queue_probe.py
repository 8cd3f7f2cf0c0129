"""How many streams a process can keep apart: the fitting step with N_EXTRA idle-but-used extra streams (each launches one tiny
kernel per step, as a communicator's stream would), under the runtime's default of 4 hardware queues and under GPU_MAX_HW_QUEUES=8.
   python tools/hw_queue_probe.py <n_extra_streams> [bind]         (set GPU_MAX_HW_QUEUES in the environment to compare)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from honerf_amd import fitting as F
n_extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device('cuda')
if len(sys.argv) > 2 and sys.argv[2] == 'bind':
    from honerf_amd.pose import bind_streams
    bind_streams(dev)            # our streams first
extra = [torch.cuda.Stream(device=dev) for _ in range(n_extra)]
junk = [torch.zeros(64, device=dev) for _ in range(n_extra)]
ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, bench.FIT_RAYS, 'f16x3', halo=True)
opt = F.make_optimizer(chain, video=False)
def step(i):
    for st, j in zip(extra, junk):
        with torch.cuda.stream(st):
            j.add_(1.0)
    F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12', pipelined=True)
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(60): step(i)
torch.cuda.synchronize()
print('GPU_MAX_HW_QUEUES=%s bind=%s extra streams %d: %.3f ms per step' % (os.environ.get('GPU_MAX_HW_QUEUES', 'default'), len(sys.argv) > 2, n_extra, (time.perf_counter() - t0) / 60 * 1e3))

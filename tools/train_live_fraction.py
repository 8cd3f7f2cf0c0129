"""Fraction of the training batch's final samples (441 rays x 128 depths, tools/train_step_bench.py's hand scene) that have a live
bone mask: what an exact far-field aggregation could remove from the training backward.  python tools/train_live_fraction.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
import torch
import train_step_bench as T
from honerf_amd import training
dev = torch.device('cuda')
ren, synth = T.build('hand', dev)
ren.precision = 'f16x3'
o, d, ex = T.rays('hand', synth, 441, dev)
out = training.render_train(ren, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], None, None, None)
z = ren.last_z_vals
dist = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1.1 / 64)], -1)
pts = o[:, None, :] + d[:, None, :] * (z + 0.5 * dist)[..., None]
cut = torch.tensor([0.08, 0.03, 0.03, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02], device=dev)
bt, tp = ex['bt_inv'], ex['T_pose']
q = torch.einsum('bij,nsj->nsbi', bt[:, :3, :3], pts) + bt[:, :3, 3] - tp
hh = 1.0 - 1.0 / (1.0 + torch.exp(-200.0 * (q.norm(dim=-1) - cut)))
live = (hh != 0).any(-1)
print('live samples %.1f %% (%d of %d)' % (100 * live.float().mean(), int(live.sum()), live.numel()))

"""One C3-style optimisation step (196 rays from a mask, 64 + 64 samples per field -> 192 shared depths, both
fields): render forward + loss terms of fitting_single.py:251-283 + backward through the render into (bt_inv, Ro, To) +
Adam step.  The pose chain that maps MANO / object parameters to (bt_inv, Ro, To) is host-side torch in the reference
(halo_util) and is not part of this path; the leaves here are the render's pose inputs themselves.
   python tools/fitting_step_bench.py [out.json]"""
import sys, os, json, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from honerf_amd import lib as L, synth
from honerf_amd.nets import SDFNetwork_OBJ, RenderingNetwork_OBJ, SingleVarianceNetwork
from honerf_amd.renderer import NeuSRenderer_fitting
from honerf_amd.fitting import render_loss_terms
lib = L.load()
dev = torch.device('cuda')
ren1, sdf, col, sc = bench.build_scene(dev, seed=9)
so, co, vo = SDFNetwork_OBJ().to(dev), RenderingNetwork_OBJ().to(dev), SingleVarianceNetwork(0.3).to(dev)
so.reset_parameters(11); co.reset_parameters(12)
ren = NeuSRenderer_fitting(sdf, ren1.deviation_network, col, so, vo, co, 64, 64, 0, 4, 1.0)
B = bench.H_IMG * bench.W_IMG
rays_o, rays_d = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
L.check(lib.hn_ray_gen(L.ptr(sc['xy']), L.ptr(sc['R']), L.ptr(sc['T']), L.ptr(sc['focal']), L.ptr(sc['principal']), 1, B,
                       L.ptr(rays_o), L.ptr(rays_d), L.stream_ptr()), 'ray_gen')
n = 196
g = torch.Generator(dev).manual_seed(3)
centre = rays_o.new_tensor([0.0, 0.0, 0.9])
bt = sc['bt_inv'].reshape(21, 4, 4).clone().requires_grad_(True)
Ro = torch.eye(3, device=dev).requires_grad_(True)
To = centre.clone().requires_grad_(True)
opt = torch.optim.Adam([bt, Ro, To], lr=1e-4)
true_rgb, true_mask = torch.rand(n, 3, device=dev, generator=g), (torch.rand(n, 1, device=dev, generator=g) > 0.3).float()

def step():
    idx = torch.randint(B // 3, 2 * B // 3, (n,), device=dev, generator=g)
    tr = torch.rand(n, 1, device=dev, generator=g)
    out = ren.render(rays_o[idx].contiguous(), rays_d[idx].contiguous(), bench.NEAR, bench.FAR, bt, sc['T_pose'], None, Ro, To, t_rand=tr)
    terms = render_loss_terms(out, true_rgb, true_mask, fit_type='12')
    opt.zero_grad(set_to_none=True)
    terms['loss'].backward()
    opt.step()
    return terms

for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K): terms = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
res = {'config': 'C3-style step: 196 rays x 192 samples, both fields, forward + loss + backward + Adam', 's_per_step': dt,
       'steps_per_s': 1.0 / dt, 'frames_per_s_at_240_steps': 1.0 / (240 * dt), 'loss': float(terms['loss'])}
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)

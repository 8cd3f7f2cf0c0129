"""Secondary measurements of SURVEY 8(d) (not the headline line of bench.py): written to a JSON file.
   * C2 frame with the reference-default 64 + 64 samples (4 importance rounds), hand nets;
   * a 512x512x(64+64) object render;
   * the forward of one C3-style fitting step: 196 rays x 192 samples through both fields (no backward yet).
   python tools/secondary_bench.py [out.json]"""
import sys, os, json, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from honerf_amd import lib as L, synth
from honerf_amd.nets import SDFNetwork, RenderingNetwork, SDFNetwork_OBJ, RenderingNetwork_OBJ, SingleVarianceNetwork
from honerf_amd.renderer import NeuSRenderer, NeuSRenderer_fitting
lib = L.load()
dev = torch.device('cuda')
out = {}

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

# ---- C2 at 64 + 64
ren, sdf, col, sc = bench.build_scene(dev, seed=9)
B = bench.H_IMG * bench.W_IMG
rays_o, rays_d = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
L.check(lib.hn_ray_gen(L.ptr(sc['xy']), L.ptr(sc['R']), L.ptr(sc['T']), L.ptr(sc['focal']), L.ptr(sc['principal']), 1, B,
                       L.ptr(rays_o), L.ptr(rays_d), L.stream_ptr()), 'ray_gen')
ren64 = NeuSRenderer(sdf, ren.deviation_network, col, 'hand', 64, 64, 0, 4, 1.0)
dt = timed(lambda: ren64.render(rays_o, rays_d, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0, t_rand=sc['t_rand']), 2)
full, sdf_only = B * 128, B * (64 + 16 * 3)
out['c2_hand_64+64'] = {'s_per_frame': dt, 'full_samples': full, 'sdf_only_samples': sdf_only,
                        'full_ray_samples_per_s': full / dt, 'all_field_evaluations_per_s': (full + sdf_only) / dt}
print('C2 hand 64+64: %.3f s per frame, %.1f M full ray-samples/s (+ %.1f M sdf-only samples per frame)' % (dt, full / dt / 1e6, sdf_only / 1e6))
# the same frame with the exact sample-level far-field skip (hn_field_set_compaction): bit-identical image
dense_img = ren64.render(rays_o, rays_d, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0, t_rand=sc['t_rand'])['color_fine'].clone()
ren64.compact_far_field = True
same = bool(torch.equal(dense_img, ren64.render(rays_o, rays_d, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0, t_rand=sc['t_rand'])['color_fine']))
dtc = timed(lambda: ren64.render(rays_o, rays_d, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, None, None, 0, t_rand=sc['t_rand']), 3)
out['c2_hand_64+64_compact'] = {'s_per_frame': dtc, 'full_ray_samples_per_s': full / dtc, 'bit_identical_to_dense': same}
print('C2 hand 64+64 with the far-field skip: %.3f s per frame, %.1f M full ray-samples/s, identical image: %s' % (dtc, full / dtc / 1e6, same))

# ---- object, 512x512, 64 + 64
so, co, vo = SDFNetwork_OBJ().to(dev), RenderingNetwork_OBJ().to(dev), SingleVarianceNetwork(0.3).to(dev)
so.reset_parameters(11); co.reset_parameters(12)
reno = NeuSRenderer(so, vo, co, 'obj', 64, 64, 0, 4, 1.0)
cam = synth.front_camera()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xy = t(synth.ndc_grid(512, 512) * 0.6)
ro, rd = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
L.check(lib.hn_ray_gen(L.ptr(xy), L.ptr(t(cam['R'])), L.ptr(t(cam['T'])), L.ptr(t(cam['focal'])), L.ptr(t(cam['principal'])), 1, B,
                       L.ptr(ro), L.ptr(rd), L.stream_ptr()), 'ray_gen')
Ro, To = torch.eye(3, device=dev), torch.zeros(3, device=dev)
dt = timed(lambda: reno.render(ro, rd, bench.NEAR, bench.FAR, None, None, None, Ro, To, 0, t_rand=sc['t_rand']), 3)
out['obj_512x512_64+64'] = {'s_per_frame': dt, 'full_ray_samples_per_s': B * 128 / dt}
print('obj 512x512 64+64: %.3f s per frame, %.1f M full ray-samples/s' % (dt, B * 128 / dt / 1e6))

# ---- one C3-style step (forward only): 196 rays, 64 + 64 per field -> 192 shared depths, both fields
renf = NeuSRenderer_fitting(sdf, ren.deviation_network, col, so, vo, co, 64, 64, 0, 4, 1.0)
n = 196
idx = torch.randint(0, B, (n,), device=dev, generator=torch.Generator(dev).manual_seed(3))
fo, fd = rays_o[idx].contiguous(), rays_d[idx].contiguous()
tr = torch.rand(n, 1, device=dev, generator=torch.Generator(dev).manual_seed(4))
To2 = torch.tensor([0.0, 0.0, 0.9], device=dev)
dt = timed(lambda: renf.render(fo, fd, bench.NEAR, bench.FAR, sc['bt_inv'], sc['T_pose'], None, Ro, To2, 0, t_rand=tr), 20)
out['c3_step_forward_196x192_dual'] = {'s_per_step': dt, 'steps_per_s': 1.0 / dt, 'note': 'the forward render of a fitting step alone; the whole step (forward + losses + backward + Adam) is bench.py fitting.single_12'}
print('C3-style step forward (196 rays x 192 samples, both fields): %.2f ms' % (dt * 1e3))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)

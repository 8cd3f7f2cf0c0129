"""Host-side time of re-packing a field inside a training loop, by stage (HN_PACK_TIMING=1 adds the library's own
stage times on stderr)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_step_bench as tb
from honerf_amd import training, nets
from honerf_amd import lib as L
dev = torch.device('cuda:0')
kind = sys.argv[1] if len(sys.argv) > 1 else 'hand'
ren, synth = tb.build(kind, dev)
ren.pack_eval_only = True
o, d, ex = tb.rays(kind, synth, 441, dev)
true_rgb, true_mask = torch.rand(441, 3, device=dev), (torch.rand(441, 1, device=dev) > 0.3).float()
opt = torch.optim.Adam(training.trainable_parameters(ren), lr=1e-4)
clk = time.perf_counter
for it in range(6):
    torch.cuda.synchronize(); t0 = clk()
    ver = nets.params_version(ren.sdf_network, ren.color_network, ren.deviation_network)
    t1 = clk()
    sd1, sd2 = ren.sdf_network.state_dict(), ren.color_network.state_dict()
    t2 = clk()
    keep = []
    d1, d2 = nets._mlp_desc(sd1, keep), nets._mlp_desc(sd2, keep)
    t3 = clk()
    new = nets.PackedField(kind, ren.sdf_network, ren.color_network, ren.deviation_network, eval_only=True)
    torch.cuda.synchronize(); t4 = clk()
    old, ren._field = ren._field, new
    ren._version = ver + (ren.precision, True)
    del old
    torch.cuda.synchronize(); t5 = clk()
    out = training.render_train(ren, o, d, 0.4, 1.5, ex['bt_inv'], ex['T_pose'], None, ex['Ro'], ex['To'])
    terms = training.train_loss(out, true_rgb, true_mask, 1.0, 1.0)
    torch.cuda.synchronize(); t6 = clk()
    opt.zero_grad(set_to_none=True)
    terms['loss'].backward()
    torch.cuda.synchronize(); t7 = clk()
    opt.step()
    torch.cuda.synchronize(); t8 = clk()
    ms = lambda a, b: (b - a) * 1e3
    print('%s step %d: version %.2f state_dict %.2f mlp_desc %.2f PackedField %.2f release old %.2f | forward %.2f backward %.2f optimiser %.2f ms'
          % (kind, it, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, t6), ms(t6, t7), ms(t7, t8)), flush=True)

"""Which synthetic scene puts samples into the CONTACT set of a fitting step (|s_h| + |s_o| < 1e-2, fitting_single.py:268-275)?
The object field is sphere-like (radius r0 in its own frame); the probe moves its centre to joint 9 + d * u for a few directions u and
distances d around r0 and counts, on the samples of one C3 step (196 rays x 192 depths) and one C5 window (4 x 40 rays), how many
fall into the contact / penetration sets.  Prints one line per candidate.  Measurement aid for tests/test_whole_step.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from honerf_amd import fitting as F  # noqa: E402


def counts(out):
    sh, so = out['sdf_hand'].reshape(-1), out['sdf_obj'].reshape(-1)
    s = sh.abs() + so.abs()
    return int((s < 1e-2).sum()), int(((sh < 0) & (so < 0)).sum()), float(sh.min()), float(sh.max()), float(so.min()), float(so.max())


def main():
    dev = torch.device('cuda')
    dirs = {'x': (1, 0, 0), 'y': (0, 1, 0), 'z': (0, 0, 1), 'xy': (0.7071, 0.7071, 0), '-y': (0, -1, 0)}
    ren, nets = bench.build_fit_nets(dev, 1, 'f16x3')
    for name, u in dirs.items():
        for d in (0.30, 0.34, 0.38, 0.40, 0.42, 0.44, 0.48):
            off = tuple(float(d * x) for x in u)
            chain, j, _ = bench.build_fit_data(dev, 40, 1, halo=True, obj_offset=off)
            views = F.synthetic_views(8, 1, bench.FIT_RAYS, 40, j[9], device=dev)
            tr = torch.rand(bench.FIT_RAYS, 1, generator=torch.Generator().manual_seed(7)).to(dev)
            tot = []
            for v in (0, 3):
                pose = chain()
                from honerf_amd import lib as L
                o, dd = F._rays(L, views[v]['xy'], views[v]['cam'], 1, bench.FIT_RAYS)
                with torch.no_grad():
                    out = ren.render(o, dd, bench.NEAR, bench.FAR, pose['bt_inv'][0], pose['T_pose_21'][0], None, pose['obj_r'][0].T.contiguous(), pose['obj_t'][0],
                                     t_rand=tr)
                tot.append(counts(out))
            print('C3 dir %-3s d %.2f  view0: contact %5d penet %5d  s_h [%.3f, %.3f] s_o [%.3f, %.3f] | view3: contact %5d penet %5d'
                  % (name, d, tot[0][0], tot[0][1], tot[0][2], tot[0][3], tot[0][4], tot[0][5], tot[1][0], tot[1][1]), flush=True)


if __name__ == '__main__':
    main()

"""Does a kernel on stream B see what a kernel on stream A wrote, when B's wait on A's event is ALREADY SATISFIED at the time B reaches
it?  (round 5: the stable term of a fitting_video step read the previous step's obj_r / obj_t / bt_inv -- same addresses every step --
when it ran on the extra stream BEHIND the pose chain's Jacobian launch, i.e. when its event wait had long been satisfied; with a
stream of its own, where the wait is a real wait, it did not.)

Stream A (the caller's) fills x with the iteration number and records an event.  Stream B first runs `busy_us` of unrelated work, then
waits for the event, then copies x -> y with (i) a torch copy (vector loads) and (ii) hn_stable_pts, the library kernel that showed it.
Counts the iterations in which y is not the iteration number."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=3000)
    ap.add_argument('--busy', type=int, default=1024, help='side of the matrix product that keeps stream B busy before its wait')
    args = ap.parse_args()
    dev = torch.device('cuda')
    from honerf_amd import lib as L
    lib = L.load()
    B = torch.cuda.Stream()
    x = torch.zeros(64, device=dev)
    y = torch.zeros(64, device=dev)
    R = torch.zeros(4, 9, device=dev)
    t = torch.zeros(4, 3, device=dev)
    pts = torch.ones(4, 40, 3, device=dev)
    pw, p0 = torch.zeros(16, 3, device=dev), torch.zeros(4, 3, device=dev)
    big = torch.randn(args.busy, args.busy, device=dev)
    bad_copy = torch.zeros(1, device=dev)
    bad_kernel = torch.zeros(1, device=dev)
    main_s = torch.cuda.current_stream()
    for k in range(1, args.iters + 1):
        x.fill_(float(k))
        R.fill_(float(k))
        t.fill_(0.0)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(B):
            if args.busy > 0:
                torch.mm(big, big)
            B.wait_event(ev)
            y.copy_(x)
            bad_copy += (y != float(k)).any().float()
            # pw[f, v] = R_f p + t_f with p = (1, 1, 1): 3 k
            L.check(lib.hn_stable_pts(L.ptr(pts), 4, 40, 10, L.ptr(R), L.ptr(t), L.ptr(pw), L.ptr(p0), L.stream_ptr()), 'hn_stable_pts')
            bad_kernel += (pw != 3.0 * k).any().float()
        main_s.wait_stream(B)       # (the next iteration's fills come after this iteration's reads)
    torch.cuda.synchronize()
    print('iterations %d, busy %d: stale torch copy in %d, stale hn_stable_pts in %d' % (args.iters, args.busy, int(bad_copy.item()), int(bad_kernel.item())))


if __name__ == '__main__':
    main()

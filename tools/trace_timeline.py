"""Timeline of one step from a rocprofv3 kernel-trace CSV: start offset, duration, stream, name (kernels > min_us shown,
runs of small ones summarised)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
step_idx = int(sys.argv[2]) if len(sys.argv) > 2 else 6
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
import re


def short(name):
    """hn:: kernels by their name; torch's element-wise kernels by the functor they apply (the template argument that says what they do)."""
    if 'at::native' in name:
        m = re.findall(r'at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]*(?:Functor|Func|Op|functor|_kernel_cuda|Kernel)[A-Za-z_0-9]*)', name)
        m = [x for x in m if x not in ('vectorized_elementwise_kernel', 'elementwise_kernel_manual_unroll', 'unrolled_elementwise_kernel', 'reduce_kernel',
                                       'ReduceOp', 'elementwise_kernel')]
        kind = 'reduce' if 'reduce_kernel' in name else 'ew'
        return 'torch %s: %s' % (kind, ','.join(dict.fromkeys(m)) if m else name.split('(')[0][:70])
    return name.split('(')[0][:70]


ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '')) for r in rows)
import os
_mark = os.environ.get('HN_TRACE_MARK')      # kernel that runs once per step (default: the hand's final evaluation)
marks = [i for i, e in enumerate(ev) if ((_mark in e[2]) if _mark else ('k_field2_hand<3>' in e[2] or 'k_field2_hand<1>' in e[2]))]
a, b = marks[step_idx], marks[step_idx + 1]
# start of step = first sdf-only hand kernel before the mark
t0 = ev[a][0]
small_n, small_t, small_start = 0, 0, None
last_end = None
for s, e, k, q in ev[a:b]:
    d = (e - s) / 1e3
    if d < min_us:
        if small_n == 0:
            small_start = s
        small_n += 1
        small_t += d
        last_small_end = e
        continue
    if small_n:
        print('  %9.1f us  [%3d small kernels, %.1f us busy, span %.1f us]' % ((small_start - t0) / 1e3, small_n, small_t, (last_small_end - small_start) / 1e3))
        small_n, small_t = 0, 0
    print('  %9.1f us  %8.1f us  q%s  %s' % ((s - t0) / 1e3, d, q, k))
if small_n:
    print('  %9.1f us  [%3d small kernels, %.1f us busy, span %.1f us]' % ((small_start - t0) / 1e3, small_n, small_t, (last_small_end - small_start) / 1e3))
print('step span %.1f us' % ((ev[b][0] - t0) / 1e3))

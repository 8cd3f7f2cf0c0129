"""Busy / idle analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals over the measured steps.
   python tools/trace_gaps.py <kernel_trace.csv> <n_steps> <skip_steps>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps, skip = int(sys.argv[2]), int(sys.argv[3])
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# find step boundaries: the forward hand full kernel appears once per step
import os
_mark = os.environ.get('HN_TRACE_MARK')      # kernel that runs once per step (default: the hand's final evaluation)
marks = [i for i, e in enumerate(ev) if ((_mark in e[2]) if _mark else ('k_field2_hand<1>' in e[2] or 'k_field2_hand<3>' in e[2]))]
marks = marks[skip:skip + steps + 1]
a, b = ev[marks[0]][0], ev[marks[-1]][0]
sel = [e for e in ev if a <= e[0] < b]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
n = len(marks) - 1
print('steps %d: span %.3f ms/step, GPU busy (union) %.3f ms/step, idle %.3f ms/step, %d dispatches/step'
      % (n, (b - a) / n / 1e6, busy / n / 1e6, (b - a - busy) / n / 1e6, len(sel) // n))
agg = {}
for s, e, k in sel:
    k = k.split('(')[0][:70]
    agg.setdefault(k, [0, 0])
    agg[k][0] += e - s
    agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
    print('  %-70s %5d/step %8.3f ms/step' % (k, c // n, t / n / 1e6))

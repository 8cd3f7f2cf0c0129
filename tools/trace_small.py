"""Small kernels of one fitting step from a rocprofv3 kernel-trace CSV, grouped by the region between the step's large
kernels: name, count, busy time.   python tools/trace_small.py kernel_trace.csv [step] [threshold_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
step_idx = int(sys.argv[2]) if len(sys.argv) > 2 else 6
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70]) for r in rows)
marks = [i for i, e in enumerate(ev) if 'k_field2_hand<3>' in e[2]]
a, b = marks[step_idx], marks[step_idx + 1]
region = collections.OrderedDict()
cur = 'after ' + ev[a][2]
for s, e, k in ev[a:b]:
    d = (e - s) / 1e3
    if d >= thr:
        cur = 'after ' + k
        continue
    region.setdefault(cur, collections.Counter())
    region[cur][k] += 1
    region[cur]['__busy_us'] += d
for name, c in region.items():
    n = sum(v for k, v in c.items() if k != '__busy_us')
    if n < 4:
        continue
    print('%s: %d small kernels, %.0f us busy' % (name, n, c['__busy_us']))
    for k, v in c.most_common(14):
        if k != '__busy_us':
            print('     %3d  %s' % (v, k))

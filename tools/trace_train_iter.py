"""One training iteration from a rocprofv3 kernel-trace CSV of tools/train_step_bench.py: every kernel with start offset, duration, queue.
   python tools/trace_train_iter.py <kernel_trace.csv> [iteration index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r.get('Queue_Id', '')) for r in rows)
marks = [i for i, e in enumerate(ev) if 'k_outer_group' in e[2]]
a, b = marks[it] + 1, marks[it + 1] + 1
t0 = ev[a][0]
for s, e, k, q in ev[a:b]:
    print('%9.1f us %8.1f us q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
print('iteration span %.1f us, %d dispatches' % ((ev[b][0] - t0) / 1e3, b - a))

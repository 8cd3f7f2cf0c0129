import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:], r.get('Queue_Id', '')) for r in rows)
# take a window in the middle: 40 ms
t_mid = ev[len(ev)//2][0]
sel = [e for e in ev if t_mid <= e[0] < t_mid + 34e6 and (e[1]-e[0]) > 150e3]
for s,e,k,q in sel:
    print('%9.1f us %8.1f us q%s %s' % ((s-t_mid)/1e3, (e-s)/1e3, q, k))
qs = sorted(set(e[3] for e in ev))
print('queues', qs)

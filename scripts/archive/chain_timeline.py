"""Kernel-by-kernel listing of the critical-path queue for a few consecutive panels of the LAST
pass in a rocprofv3 kernel trace of bench.py: start offset, duration, gap to the previous kernel
on the same queue, grid size."""
import csv, sys, collections
f = sys.argv[1]
first, count = int(sys.argv[2]), int(sys.argv[3])
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
grams = [i for i, r in enumerate(tr) if 'gram_kernel' in r['Kernel_Name']]
tr2 = tr[grams[-2]:]
t0 = int(tr2[0]['Start_Timestamp'])
qs = collections.Counter(r['Queue_Id'] for r in tr2)
print('queues', dict(qs), 'span %.2f ms' % ((max(int(r['End_Timestamp']) for r in tr2) - t0) / 1e6))
for q in qs:
    rows = [r for r in tr2 if r['Queue_Id'] == q]
    print('== queue', q)
    prev_end = None
    for r in rows[first:first + count]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        n = r['Kernel_Name']
        name = n.split('(')[0].replace('void ', '')[:44]
        blocks = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
        print('%9.1f us  dur %7.1f  gap %6.1f  blocks %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0, blocks, name))
        prev_end = e

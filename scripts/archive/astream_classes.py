"""Critical-stream kernel classes of one pass of a rocprofv3 kernel trace (development aid):
usage: python scripts/astream_classes.py <kernel_trace.csv> [pass_index]"""
import csv, re, sys
from collections import defaultdict
def short(name):
    m = re.search(r'gemm_nt_kernel<(\w+), (\d+), (\d+), \d+, \d+, (\d+)', name)
    if m: return 'gemm%sx%s_s%s' % (m.group(2), m.group(3), m.group(4))
    m = re.search(r'(\w+)<', name); return m.group(1) if m else name[:20]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
grams = [i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name']]
gmax = max(int(rows[i]['Grid_Size_X']) for i in grams)
starts = [i for i in grams if int(rows[i]['Grid_Size_X']) == gmax]
ps = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
lo = starts[ps]; hi = starts[ps + 1] if ps + 1 < len(starts) else len(rows)
ev = rows[lo:hi]
wall = (max(int(r['End_Timestamp']) for r in ev) - int(ev[0]['Start_Timestamp'])) / 1e6
cnt = defaultdict(int)
for r in ev: cnt[r['Queue_Id']] += 1
A = max(cnt, key=cnt.get)
agg = defaultdict(lambda: [0, 0.0, 0.0])
for r in ev:
    wgs = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    cls = ('A ' if r['Queue_Id'] == A else 'B ') + short(r['Kernel_Name']) + (' tiny' if wgs <= 16 else (' mid' if wgs <= 1024 else ' big'))
    a = agg[cls]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
print('wall %.3f ms' % wall)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print('   %-30s n=%4d total %8.2f ms avg %7.1f us max %7.1f us' % (k, v[0], v[1] / 1e3, v[1] / v[0], v[2]))

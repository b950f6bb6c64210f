"""Per-queue (stream) breakdown of the LAST pass in a rocprofv3 kernel trace of bench.py."""
import csv, glob, collections, sys
f = sys.argv[1]
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
grams = [i for i, r in enumerate(tr) if 'gram_kernel' in r['Kernel_Name']]
tr2 = tr[grams[-2]:]
t0 = int(tr2[0]['Start_Timestamp'])
print('pass span %.1f ms' % ((max(int(r['End_Timestamp']) for r in tr2) - t0) / 1e6))
qs = collections.Counter(r['Queue_Id'] for r in tr2)
for q, cnt in qs.items():
    rows = [r for r in tr2 if r['Queue_Id'] == q]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    print('queue', q, 'kernels', cnt, 'busy %.1f ms' % (busy / 1e6), 'first %.1f last %.1f' % ((int(rows[0]['Start_Timestamp']) - t0) / 1e6, (max(int(r['End_Timestamp']) for r in rows) - t0) / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        n = r['Kernel_Name']; d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        if 'gemm_nt' in n:
            cfg = n.split('<')[1].split('>')[0].replace('double, ', '')
            blocks = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
            key = (cfg, '<=4' if blocks <= 4 else '<=64' if blocks <= 64 else '<=1024' if blocks <= 1024 else 'big')
        else:
            key = (n.split('(')[0][5:30], '')
        agg[key][0] += 1; agg[key][1] += d
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print('    ', k, v[0], '%.1f ms' % (v[1] / 1e6), 'avg %.1f us' % (v[1] / v[0] / 1e3))

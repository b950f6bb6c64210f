"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals, GEMM time by tile config / grid."""
import csv, collections, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/**/*_kernel_trace.csv', recursive=True))[-1]
tr = list(csv.DictReader(open(f)))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = collections.defaultdict(lambda: [0, 0])
agg = collections.defaultdict(lambda: [0, 0])
for r in tr:
    n = r['Kernel_Name']; d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    short = n.split('(')[0].replace('void ', '')
    tot[short][0] += 1; tot[short][1] += d
    if 'gemm_nt' in n:
        cfg = n.split('<')[1].split('>')[0]
        wx = int(r['Workgroup_Size_X']); gx = int(r['Grid_Size_X']) // wx; gy = int(r['Grid_Size_Y'])
        agg[(cfg, gx, gy)][0] += 1; agg[(cfg, gx, gy)][1] += d
tr.sort(key=lambda r: int(r['Start_Timestamp']))
span = (max(int(r['End_Timestamp']) for r in tr) - int(tr[0]['Start_Timestamp'])) / 1e6
busy = sum(v[1] for v in tot.values()) / 1e6
print('kernels %d span %.1f ms busy %.1f ms (per pass: /%.0f)' % (len(tr), span, busy, div))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print('%-60s calls %6d total %8.2f ms avg %9.1f us' % (k[:60], v[0] / div, v[1] / 1e6 / div, v[1] / v[0] / 1e3))
print('--- GEMM by (cfg, grid)')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(k, 'calls', v[0] / div, 'total ms %.2f' % (v[1] / 1e6 / div), 'avg us %.1f' % (v[1] / v[0] / 1e3))

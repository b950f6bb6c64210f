"""Per-launch table (m, n, K, tiles, duration, TFLOP/s) of the MFMA GEMM and stripe-solve launches of ONE pass of
bench.py: joins a rocprofv3 kernel trace with the library's launch log (G3_GEMM_LOG=<file>, one line per launch in host
order = dispatch order).  usage: python scripts/launch_table.py <kernel_trace.csv> <gemm.log> <passes in the trace> [title]"""
import csv, sys, collections
trace, log, passes = sys.argv[1], sys.argv[2], int(sys.argv[3])
title = sys.argv[4] if len(sys.argv) > 4 else ''
rows = [r for r in csv.DictReader(open(trace)) if 'gemm_nt_kernel' in r['Kernel_Name'] or 'trsm_stripe_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
lines = [l.split() for l in open(log) if l.strip()]
assert len(rows) == len(lines), (len(rows), len(lines))
per = len(rows) // passes
rows, lines = rows[-per:], lines[-per:]
t0 = min(int(r['Start_Timestamp']) for r in rows)
print('# Per-launch table%s' % ((': ' + title) if title else ''))
print()
print('Last of %d passes in the trace; launches in dispatch order.  `stream` 1 = the low-priority bulk stream.  TFLOP/s counts the')
print('algorithmic flops of the wanted elements (2 k per element; the stripe solve m n^2) over the launch\'s duration IN THE SWEEP,')
print('i.e. while kernels of the other stream share the chip.')
print()
print('| # | start ms | kernel | stream | m | n | K | shape | tiles | us | TFLOP/s |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, (r, l) in enumerate(zip(rows, lines)):
    kind, bm, bn, waves, m, n, k, shp, tiles, flops, side = l[0], int(l[1]), int(l[2]), int(l[3]), int(l[4]), int(l[5]), int(l[6]), int(l[7]), int(l[8]), float(l[9]), int(l[10])
    assert (kind == 'trsm') == ('trsm_stripe' in r['Kernel_Name']), (i, l, r['Kernel_Name'])
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = ('gemm %dx%d/%dw' % (bm, bn, waves)) if kind == 'gemm' else 'trsm stripe %d' % bm
    shape = {0: 'dense', 1: 'lower', 2: 'stair'}[shp]
    print('| %d | %.3f | %s | %d | %d | %d | %d | %s | %d | %.1f | %.1f |' % (i, (int(r['Start_Timestamp']) - t0) / 1e6, name, side, m, n, k, shape, tiles, dur, flops / dur / 1e6))
    a = agg[(name, side, k)]
    a[0] += 1; a[1] += dur; a[2] += flops
print()
print('| kernel | stream | K | launches | total us | flops | TFLOP/s in the sweep |')
print('|---|---|---|---|---|---|---|')
for (name, side, k), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('| %s | %d | %d | %d | %.0f | %.3e | %.1f |' % (name, side, k, a[0], a[1], a[2], a[2] / a[1] / 1e6))

"""gaps between consecutive big-tile GEMM launches in a rocprofv3 kernel trace (last evaluation in the file):
where the bulk stream waited for the chain.  usage: r4_bulk_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last evaluation: from the last gram_kernel burst on
gi = [i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name']]
# first gram of the last burst
k = gi[-1]
while k - 1 in gi or (k > 0 and any(j in gi for j in range(k - 40, k))):
    k = max(j for j in gi if j < k)
    if not any(j in gi for j in range(k - 40, k)): break
ev = rows[k:]
t0 = int(ev[0]['Start_Timestamp'])
big = [r for r in ev if 'gemm_nt_kernel<double, 128, 128' in r['Kernel_Name']]
print('evaluation span %.2f ms, %d kernels, %d big GEMM launches' % ((int(ev[-1]['End_Timestamp']) - t0) / 1e6, len(ev), len(big)))
prev_end = None
tot_gap = 0
for r in big:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    if prev_end: tot_gap += max(gap, 0)
    print('start %8.2f ms  dur %8.3f ms  gap before %8.1f us  grid %s' % ((s - t0) / 1e6, (e - s) / 1e6, gap, r['Grid_Size_X']))
    prev_end = e if prev_end is None or e > prev_end else prev_end
print('sum of gaps %.2f ms; last big GEMM ends at %.2f ms' % (tot_gap / 1e3, (prev_end - t0) / 1e6))

"""Timeline analysis of a rocprofv3 --kernel-trace CSV (development aid).
usage: python scripts/trace_timeline.py <kernel_trace.csv> [pass_index]
Splits the trace into passes at each gram_kernel<..., true/false> launch with the big grid (the
start of a step), then prints per-stream busy time, the wall time of the pass, the total kernel
time by kernel family, and (with -v) the first events of the pass."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'gemm_nt_kernel<(\w+), (\d+), (\d+)', name)
    if m:
        return 'gemm%sx%s' % (m.group(2), m.group(3))
    m = re.search(r'(\w+)<', name)
    if m:
        return m.group(1)
    return name[:30]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # passes start at the symmetric Gram launch (the largest gram_kernel grid)
    grams = [i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name']]
    gmax = max(int(rows[i]['Grid_Size_X']) * int(rows[i]['Grid_Size_Y']) for i in grams)
    starts = [i for i in grams if int(rows[i]['Grid_Size_X']) * int(rows[i]['Grid_Size_Y']) == gmax]
    which = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(starts) - 1
    lo = starts[which]
    hi = starts[which + 1] if which + 1 < len(starts) else len(rows)
    ev = rows[lo:hi]
    t0 = int(ev[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in ev)
    print('pass %d of %d: %d kernels, wall %.3f ms' % (which, len(starts), len(ev), (t1 - t0) / 1e6))
    by_stream = defaultdict(list)
    fam = defaultdict(lambda: [0, 0.0])
    for r in ev:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        by_stream[r['Queue_Id'] + '/' + r['Stream_Id']].append((s, e, short(r['Kernel_Name']), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
        f = fam[short(r['Kernel_Name'])]
        f[0] += 1
        f[1] += (e - s) / 1e6
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print('  %-24s n=%5d  total %8.3f ms  avg %8.1f us' % (k, v[0], v[1], v[1] / v[0] * 1e3))
    for sid, lst in by_stream.items():
        busy = sum(e - s for s, e, _, _ in lst) / 1e6
        gaps = sum(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)) / 1e6
        print('stream %s: %d kernels, busy %.3f ms, gaps between kernels %.3f ms, first %.3f last %.3f'
              % (sid, len(lst), busy, gaps, lst[0][0] / 1e6, lst[-1][1] / 1e6))
    if '-v' in sys.argv:
        n = int(sys.argv[sys.argv.index('-v') + 1]) if sys.argv.index('-v') + 1 < len(sys.argv) else 80
        allev = sorted((s, e, sid, nm, g) for sid, lst in by_stream.items() for s, e, nm, g in lst)
        for s, e, sid, nm, g in allev[:n]:
            print('  %9.1f us  +%7.1f us  %-8s %-14s wgs=%d' % (s / 1e3, (e - s) / 1e3, sid, nm, g))


if __name__ == '__main__':
    main()

"""Sum a rocprofv3 --pmc counter per kernel family from counter_collection.csv files under a directory.
usage: python scripts/pmc_sum.py <dir> [substring-of-kernel-name]"""
import csv, glob, sys, re
from collections import defaultdict
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if pat not in k:
            continue
        m = re.search(r'gemm_nt_kernel<(\w+), (\d+), (\d+), \d+, \d+, (\d+)', k)
        key = ('gemm %sx%s st%s' % (m.group(2), m.group(3), m.group(4))) if m else k[:40]
        grid = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
        key += ' grid>=%d' % (1 << (grid.bit_length() - 1)) if grid else ''
        acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[key].add(r['Dispatch_Id'])
for k in sorted(acc):
    n = len(cnt[k])
    print('%-48s dispatches %5d  ' % (k, n) + '  '.join('%s/dispatch = %.4g' % (c, v / n) for c, v in acc[k].items()))

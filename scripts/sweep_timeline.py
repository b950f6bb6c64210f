import csv, sys, re, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last sweep: from the last g3_gram_jit launch
idx = [i for i, r in enumerate(rows) if 'gram' in r['Kernel_Name']]
lo = idx[-1]
ev = rows[lo:]
t0 = int(ev[0]['Start_Timestamp'])
def short(n):
    m = re.search(r'gemm_nt_kernel<\w+, (\d+), (\d+)', n)
    if m: return 'gemm%sx%s' % (m.group(1), m.group(2))
    m = re.search(r'(\w+)<', n)
    return (m.group(1) if m else n)[:28]
for r in ev:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print('%8.1f %8.1f %7.1f us  q%s  %-28s grid %s x %s / wg %s' % (s, e, e - s, r['Queue_Id'], short(r['Kernel_Name']), r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X']))

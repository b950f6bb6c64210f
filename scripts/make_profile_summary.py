"""Turn the rocprofv3 outputs of one round (gpurun_out/rNN/{trace,pmc_fetch,pmc_write}) into the
committed summaries under profiles/: kernel stats CSV, a markdown summary and the HBM-traffic
JSON that bench.py reports as roofline.traffic.
usage: python scripts/make_profile_summary.py r02 [passes_in_trace] [trace_subdir] [out_suffix] [command text]
  e.g.  make_profile_summary.py r02 4                       -> profiles/r02_summary.md (+ traffic)
        make_profile_summary.py r02 7 c2 _c2 "bench.py --points 8192 ..."   -> profiles/r02_c2_summary.md"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
sub = sys.argv[3] if len(sys.argv) > 3 else 'trace'
suffix = sys.argv[4] if len(sys.argv) > 4 else ''
command = sys.argv[5] if len(sys.argv) > 5 else 'python3 bench.py --steps 3 --warmup 1 --cpu-n 0'
src = os.path.join('gpurun_out', tag)
os.makedirs('profiles', exist_ok=True)
stats = glob.glob(os.path.join(src, sub, '*', '*_kernel_stats.csv'))[0]
shutil.copy(stats, 'profiles/%s%s_kernel_stats.csv' % (tag, suffix))
trace = glob.glob(os.path.join(src, sub, '*', '*_kernel_trace.csv'))[0]
tr = list(csv.DictReader(open(trace)))

def klass(name, grid, wg):
    if 'gemm_nt' in name:
        cfg = name.split('<')[1].split('>')[0].replace(' ', '')
        blocks = int(grid) // int(wg)
        if ',128,128,64,' in cfg:
            return 'gemm_nt<%s> grid>=4096 (bulk panel updates)' % cfg if blocks >= 4096 else 'gemm_nt<%s> grid<4096' % cfg
        return 'gemm_nt<%s>' % cfg
    return name.split('(')[0].replace('void ', '')

agg = collections.defaultdict(lambda: [0, 0])
for r in tr:
    k = klass(r['Kernel_Name'], r['Grid_Size_X'], r['Workgroup_Size_X'])
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
lines = ['# rocprofv3 summary %s%s' % (tag, suffix), '',
         'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- %s`' % command,
         '(1x MI355X; %d passes of the hot path in the trace, warm-up included).' % passes,
         'Raw per-kernel stats: `%s%s_kernel_stats.csv`.  Kernels on the two look-ahead streams overlap, so' % (tag, suffix),
         'summed kernel time exceeds wall time.', '',
         '| kernel (split by grid size) | launches/pass | total ms/pass | avg us/launch |', '|---|---|---|---|']
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    lines.append('| %s | %.1f | %.2f | %.1f |' % (k, v[0] / passes, v[1] / 1e6 / passes, v[1] / v[0] / 1e3))
traffic = {}
for name, counter in [] if suffix else (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    fs = glob.glob(os.path.join(src, name, '*', '*_counter_collection.csv'))
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = klass(r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'])
        a[k][0] += 1
        a[k][1] += float(r['Counter_Value'])
    traffic[counter] = {k: {'launches': v[0], 'sum_kb': v[1]} for k, v in a.items()}
if traffic:
    bulk = [k for k in traffic['FETCH_SIZE'] if 'bulk' in k][0]
    f, w = traffic['FETCH_SIZE'][bulk], traffic['WRITE_SIZE'][bulk]
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KB) reads exactly half of a wide coalesced stream on
    # gfx950 -> x2; WRITE_SIZE (KB) is exact (checked here on the Gram kernel: known bytes written).
    fetch_b = 2.0 * f['sum_kb'] * 1024 / f['launches']
    write_b = w['sum_kb'] * 1024 / w['launches']
    gram = [k for k in traffic['WRITE_SIZE'] if 'gram_kernel' in k]
    cal = traffic['WRITE_SIZE'][gram[0]]['sum_kb'] * 1024 if gram else None
    out = {'kernel': bulk, 'launches': f['launches'], 'fetch_bytes_per_launch': fetch_b, 'write_bytes_per_launch': write_b,
           'hbm_bytes_per_launch': fetch_b + write_b, 'gram_write_bytes_measured': cal,
           'gram_write_bytes_expected': (512 * 513 // 2) * 64 * 64 * 8 + 1024 * 32768 * 8,
           'correction': 'FETCH_SIZE x2 (gfx950 wide-stream undercount), WRITE_SIZE x1; separate --pmc passes'}
    json.dump(out, open('profiles/%s_traffic.json' % tag, 'w'), indent=1)
    lines += ['', '## HBM traffic of the bulk GEMM launches (PMC, separate passes)', '',
              '`rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` on `python3 bench.py --steps 1 --warmup 0 --cpu-n 0 --skip-events`.',
              'Per launch (average over %d launches): fetch %.3f GB (FETCH_SIZE x 2, gfx950 correction), write %.3f GB.' % (f['launches'], fetch_b / 1e9, write_b / 1e9),
              'Calibration on a known byte count: the Gram kernels wrote %.4f GB by WRITE_SIZE vs %.4f GB expected.' % ((cal or 0) / 1e9, out['gram_write_bytes_expected'] / 1e9)]
open('profiles/%s%s_summary.md' % (tag, suffix), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

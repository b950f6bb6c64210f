"""profiles/rNN_mfma.md from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass over bench.py
(kernels run serialised under counter collection, so durations here are stand-alone, not in-sweep).
usage: python scripts/mfma_util_summary.py gpurun_out/pmc_mfma r01"""
import collections, csv, glob, sys
src, tag = sys.argv[1], sys.argv[2]
f = glob.glob(src + '/*/*counter_collection.csv')[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = by[r['Dispatch_Id']]
    d[r['Counter_Name']] = float(r['Counter_Value'])
    d['name'], d['grid'], d['wg'] = r['Kernel_Name'], int(r['Grid_Size']), int(r['Workgroup_Size'])
    d['dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d in by.values():
    n = d['name']
    if 'gemm_nt' in n:
        cfg = n.split('<')[1].split('>')[0].replace(' ', '')
        key = 'gemm_nt<%s>' % cfg + (' grid>=4096 (bulk panel updates)' if ',128,128,64,' in cfg and d['grid'] // d['wg'] >= 4096 else '')
    else:
        key = n.split('(')[0].replace('void ', '')
    a = agg[key]
    a[0] += 1; a[1] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0); a[2] += d.get('GRBM_GUI_ACTIVE', 0); a[3] += d['dur']
NXCD, NSIMD = 8, 1024
lines = ['# MFMA utilisation %s (PMC)' % tag, '',
         '`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --cpu-n 0 --skip-events`',
         '(N=32768, fp64; kernels are serialised under counter collection: stand-alone durations).',
         'GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GUI_ACTIVE / 8 / duration.  SQ_VALU_MFMA_BUSY_CYCLES is',
         'summed over the 1024 SIMDs (64 cycles per v_mfma_f64_16x16x4_f64): utilisation = busy / 1024 / (GUI_ACTIVE / 8).', '',
         '| kernel | launches | total ms | shader clock GHz | MFMA pipe busy |', '|---|---|---|---|---|']
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][3])[:6]:
    cyc = a[2] / NXCD
    clock = '%.2f' % (cyc / a[3]) if a[3] > 50e6 else 'n/a (short launches)'
    lines.append('| %s | %d | %.1f | %s | %.1f %% |' % (k, a[0], a[3] / 1e6, clock, 100.0 * a[1] / NSIMD / max(cyc, 1)))
bulk = [a for k, a in agg.items() if 'bulk' in k][0]
clk = bulk[2] / NXCD / bulk[3]
lines += ['', 'The FP64 matrix peak of 78.6 TFLOP/s assumes 2.4 GHz; under sustained FP64 MFMA load the shader clock is',
          '%.2f GHz, i.e. a clock-adjusted peak of %.1f TFLOP/s.' % (clk, 78.6 * clk / 2.4)]
open('profiles/%s_mfma.md' % tag, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

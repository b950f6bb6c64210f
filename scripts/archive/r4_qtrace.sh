#!/bin/bash
# classic vs chain sweep, same box: per-queue tables and the bulk stream's launches side by side
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
G3_CHAIN=$mode G3_CHAIN_WGS=32 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/m$mode -- python3 $R/bench.py --cpu-n 0 --no-api --no-measure-traffic --skip-events --points 8192 --steps 3 --warmup 1 > $OUT/m$mode.log 2>&1; echo "trace $mode rc=$?"
done
cd $R
for mode in 0 1; do
f=$(ls $OUT/m$mode/*/*kernel_trace.csv | head -1)
python - "$f" > $OUT/bulk$mode.txt <<'PY'
import csv, sys, collections
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
grams = [i for i, r in enumerate(tr) if 'gram_kernel' in r['Kernel_Name']]
tr2 = tr[grams[-2]:]
t0 = int(tr2[0]['Start_Timestamp'])
print('pass span %.2f ms' % ((max(int(r['End_Timestamp']) for r in tr2) - t0) / 1e6))
big = [r for r in tr2 if 'gemm_nt' in r['Kernel_Name'] and int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) > 2000]
qb = collections.Counter(r['Queue_Id'] for r in big).most_common(1)[0][0]
print('bulk queue', qb, ' columns:', [k for k in tr2[0].keys()][:20])
for r in tr2:
    if r['Queue_Id'] != qb: continue
    print('%9.1f %8.1f  grid %7d wg %4d lds %6s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
          int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Workgroup_Size_X']), r.get('LDS_Block_Size', r.get('Lds_Block_Size', '?')), r['Kernel_Name'][:40]))
PY
done

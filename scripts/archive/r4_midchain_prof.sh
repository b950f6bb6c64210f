#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/midchain; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for N in ${NS:-1024 2048}; do
rm -rf $OUT/t$N
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$N -- python3 $R/scripts/r4_midchain_one.py $N 96 2>&1 | grep "N="
f=$(ls $OUT/t$N/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:12]:
    print('%-86s calls %5s total %8.3f ms avg %8.1f us' % (r['Name'][:86], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print('sum %.3f ms over 7 sweeps' % (tot / 1e6))
PY
done

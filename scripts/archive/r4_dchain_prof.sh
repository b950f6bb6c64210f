#!/bin/bash
# kernel-level profile of dlogp_chain at N = 128, 4096 rows
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dchain; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/scripts/r4_dchain_one.py > $OUT/log.txt 2>&1; echo rc=$?
cd $R
f=$(ls $OUT/t/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('%-100s calls %6s total %9.3f ms avg %9.1f us' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY

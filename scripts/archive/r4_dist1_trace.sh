#!/bin/bash
# kernel-class totals of one evaluation: in-library sweep vs the native multi-GPU driver at one rank
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dist1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --cpu-n 0 --no-api --no-measure-traffic --skip-events --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one -- python3 $B > $OUT/one.log 2>&1; echo "one rc=$?"
export G3_FORCE_DIST=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dist -- python3 $B > $OUT/dist.log 2>&1; echo "dist rc=$?"
cd $R
for w in one dist; do
  f=$(ls $OUT/$w/*/*kernel_stats.csv | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('%-90s calls %6s total %9.3f ms avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print('sum over kernels %.3f ms (4 evaluations)' % (tot / 1e6))
PY
done

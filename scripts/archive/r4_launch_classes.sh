#!/bin/bash
# launch classes of the headline step in the sweep: default schedule, and with the tall solve split at the launch level
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04lc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --cpu-n 0 --no-api --no-measure-traffic --skip-events"
rm -f $OUT/c4_gemm.log $OUT/c4s_gemm.log
G3_GEMM_LOG=$OUT/c4_gemm.log timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c4 -- python3 $B --steps 2 --warmup 1 > $OUT/c4.log 2>&1; echo "c4 rc=$?"
G3_TRSM_SPLIT_MIN=8192 G3_TRSM_SPLIT_N=256 G3_GEMM_LOG=$OUT/c4s_gemm.log timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c4s -- python3 $B --steps 2 --warmup 1 > $OUT/c4s.log 2>&1; echo "c4 split rc=$?"
grep -h "^{" $OUT/c4.log $OUT/c4s.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ms_per_step %.2f' % d['ms_per_step'])"

#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03lt}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --cpu-n 0 --no-api --no-measure-traffic --skip-events"
G3_GEMM_LOG=$OUT/c2_gemm.log timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c2 -- python3 $B --points 8192 --steps 3 --warmup 1 > $OUT/c2.log 2>&1; echo "c2 rc=$?"
G3_GEMM_LOG=$OUT/c3_gemm.log timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c3 -- python3 $B --points 16384 --dims 8 --kernel mat52cos --steps 3 --warmup 1 > $OUT/c3.log 2>&1; echo "c3 rc=$?"
G3_GEMM_LOG=$OUT/c4_gemm.log timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c4 -- python3 $B --steps 2 --warmup 1 > $OUT/c4.log 2>&1; echo "c4 rc=$?"

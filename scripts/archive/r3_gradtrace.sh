#!/bin/bash
# kernel trace of the distributed gradient (one rank through RCCL, N=32768): which kernels carry the gradient mode
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3gt}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export G3_FORCE_DIST=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-n 0 --no-api --no-measure-traffic --grad --steps 1 --warmup 1 --skip-events > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $R
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv; head -25 $OUT/kernel_stats.csv | cut -c1-160

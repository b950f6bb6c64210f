#!/bin/bash
OUT=gpurun_out/${1:-r3e}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_distributed.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api"
echo "== native driver, one rank, RCCL" | tee -a $OUT/dist.log
G3_FORCE_DIST=1 timeout -k 10 300 $B --steps 4 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_dist1_native_rccl.json | cut -c1-200 | tee -a $OUT/dist.log
echo "== python driver, one rank, nccl" | tee -a $OUT/dist.log
G3_FORCE_DIST=1 G3_DIST_DRIVER=python G3_DIST_COLLECTIVES=1 timeout -k 10 300 $B --steps 4 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_dist1_python_nccl.json | cut -c1-200 | tee -a $OUT/dist.log
cd /tmp && export TMPDIR=/tmp
G3_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_native -- python3 $GRAFT_REPO_ROOT/bench.py --no-measure-traffic --cpu-n 0 --no-api --steps 2 --warmup 1 --skip-events > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
echo trace rc=$?

#!/bin/bash
# round-3 distributed checks on ONE GPU: the native driver through RCCL with one rank, the python driver through
# ProcessGroupNCCL with one rank, two-rank rehearsals over gloo of both drivers
OUT=gpurun_out/${1:-r3d}; mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api"
echo "== single GPU in-library" | tee -a $OUT/dist.log
timeout -k 10 300 $B --steps 4 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_1gpu.json | cut -c1-400 | tee -a $OUT/dist.log
echo "== native driver, one rank, RCCL" | tee -a $OUT/dist.log
G3_FORCE_DIST=1 timeout -k 10 300 $B --steps 4 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_dist1_native_rccl.json | cut -c1-600 | tee -a $OUT/dist.log
echo "== python driver, one rank, ProcessGroupNCCL (forced collectives)" | tee -a $OUT/dist.log
G3_FORCE_DIST=1 G3_DIST_DRIVER=python G3_DIST_COLLECTIVES=1 timeout -k 10 300 $B --steps 4 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_dist1_python_nccl.json | cut -c1-600 | tee -a $OUT/dist.log
echo "== native driver, nb 512 one rank" | tee -a $OUT/dist.log
G3_FORCE_DIST=1 timeout -k 10 300 $B --steps 4 --warmup 1 --panel 512 2>>$OUT/dist.err | tee $OUT/bench_dist1_native_nb512.json | cut -c1-300 | tee -a $OUT/dist.log
echo "== two ranks on one GPU, native driver, callbacks over gloo (N=16384)" | tee -a $OUT/dist.log
G3_DIST_BACKEND=gloo timeout -k 10 400 $B --gpus 2 --points 16384 --steps 2 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_dist2_native_gloo.json | cut -c1-600 | tee -a $OUT/dist.log
echo "== two ranks on one GPU, python driver over gloo (N=16384)" | tee -a $OUT/dist.log
G3_DIST_BACKEND=gloo G3_DIST_DRIVER=python timeout -k 10 400 $B --gpus 2 --points 16384 --steps 2 --warmup 1 2>>$OUT/dist.err | tee $OUT/bench_dist2_python_gloo.json | cut -c1-600 | tee -a $OUT/dist.log
tail -5 $OUT/dist.err

#!/bin/bash
# round-3 repeatability soaks with the eight-wave bulk tile, the reordered K loop and the new Gram paths
OUT=gpurun_out/${1:-r3soak}; mkdir -p $OUT
(echo "== potrf soak fp64"; timeout -k 10 500 python scripts/potrf_soak.py 200 3;
 echo "== potrf soak fp32"; timeout -k 10 300 python scripts/potrf_soak.py 80 3 f32;
 echo "== sweep soak"; timeout -k 10 600 python scripts/sweep_soak.py 3072 6144 8192 12288 16384;
 echo "== jitter accuracy"; timeout -k 10 300 python scripts/jitter_accuracy.py) 2>&1 | grep -v amdgpu | tee $OUT/soak.log
timeout -k 10 400 python -m pytest tests/test_gpu_distributed.py -x -q -k "native_driver_matches" 2>&1 | tail -2 | tee -a $OUT/soak.log

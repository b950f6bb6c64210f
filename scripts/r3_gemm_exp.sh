#!/bin/bash
# (needs scripts/build_variant.sh exp g3_gemm.hip -DG3_GEMM_EXPERIMENTS in the build container)
OUT=gpurun_out/${1:-r3exp}; mkdir -p $OUT
S="30720 30720 1024 1  30720 30720 2048 1  30720 1024 1024 0  16384 1024 1024 0  8192 1024 1024 0 16384 16384 512 1 8192 8192 512 1 6144 512 512 0 4096 4096 256 1"
for cfg in 5 6 7 8 9 3; do
  echo "== G3_GEMM_CFG=$cfg" | tee -a $OUT/exp.log
  G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_exp.so G3_GEMM_CFG=$cfg timeout -k 10 200 python scripts/gemm_bench.py $S 2>&1 | grep -v amdgpu | tee -a $OUT/exp.log
done

#!/bin/bash
# (needs the A/B build: scripts/build_variant.sh order0 g3_gemm.hip -DG3_GEMM_ORDER=0, in the build container)
# stand-alone A/B of GEMM tile variants on the shapes of the headline sweep
OUT=gpurun_out/${1:-r3ab}; mkdir -p $OUT
S="30720 30720 1024 1  30720 30720 2048 1  30720 1024 1024 0  16384 1024 1024 0  8192 1024 1024 0 16384 16384 512 1 8192 8192 512 1 6144 512 512 0"
for cfg in 0 2 5 3; do
  echo "== G3_GEMM_CFG=$cfg" | tee -a $OUT/ab.log
  G3_GEMM_CFG=$cfg timeout -k 10 200 python scripts/gemm_bench.py $S 2>&1 | tee -a $OUT/ab.log
done
echo "== order0 lib" | tee -a $OUT/ab.log
G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_order0.so timeout -k 10 200 python scripts/gemm_bench.py $S 2>&1 | tee -a $OUT/ab.log

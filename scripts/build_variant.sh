#!/bin/bash
# Build a measurement variant of the library next to the product build (never loaded unless G3_LIB_PATH names it):
#   scripts/build_variant.sh ctrace g3_potrf.hip -DG3_CHAIN_TRACE     -> g3py_amd/lib/libg3hip_ctrace.so
#   scripts/build_variant.sh probe g3_potrf.hip -DG3_PROBE
# usage: build_variant.sh <name> <source file to recompile> <extra hipcc flags...>
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/g3py_amd/csrc -j4 > /dev/null
obj=/tmp/g3_variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -Wall -Wno-unused-function -Wno-pass-failed "$@" -c $R/g3py_amd/csrc/$src -o $obj
objs=""
for f in g3_gemm g3_potrf g3_gram g3_gram_jit g3_grad g3_api g3_dist; do
  if [ "$f.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs $R/g3py_amd/lib/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/g3py_amd/lib/libg3hip_$name.so $objs -ldl
echo $R/g3py_amd/lib/libg3hip_$name.so

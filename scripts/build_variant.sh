#!/bin/bash
# Build a measurement variant of the library next to the product build (never loaded unless G3_LIB_PATH names it):
#   scripts/build_variant.sh probe g3_potrf.hip -DG3_PROBE                 -> g3py_amd/lib/libg3hip_probe.so
#   scripts/build_variant.sh chain all -DG3_CHAIN_SERVER                   -> g3py_amd/lib/libg3hip_chain.so
# usage: build_variant.sh <name> <source file to recompile | all> <extra hipcc flags...>
# (`all` recompiles every translation unit: needed when the flags change a shared struct, e.g. G3_CHAIN_SERVER adds
#  fields to the context.)  Variant .so files are scratch: `make clean` removes them and .gpurunignore keeps them off the GPU box
# unless a script asks for them explicitly.
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/g3py_amd/csrc -j4 > /dev/null
objs=""
for f in g3_gemm g3_potrf g3_gram g3_gram_jit g3_grad g3_api g3_dist g3_chainb; do
  [ -f $R/g3py_amd/csrc/$f.hip ] || continue
  if [ "$f.hip" == "$src" ] || [ "$src" == "all" ]; then
    obj=/tmp/g3_variant_${name}_$f.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -Wall -Wno-unused-function -Wno-pass-failed "$@" -c $R/g3py_amd/csrc/$f.hip -o $obj
    objs="$objs $obj"
  else
    objs="$objs $R/g3py_amd/lib/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/g3py_amd/lib/libg3hip_$name.so $objs -ldl
echo $R/g3py_amd/lib/libg3hip_$name.so

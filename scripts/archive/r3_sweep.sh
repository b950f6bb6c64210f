#!/bin/bash
# (needs the A/B build: scripts/build_variant.sh order0 g3_gemm.hip -DG3_GEMM_ORDER=0, in the build container)
# round-3 measurement sweep (one gpurun call): parity suite, headline bench, super-panel / panel-width sweep at
# configs 2-4, A/B of the K-loop order.  Everything goes to gpurun_out/$1/
set -o pipefail
OUT=gpurun_out/${1:-r3a}
mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api"
run() { # name, env..., -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "== $name" >> $OUT/sweep.log
  env "${envs[@]}" timeout -k 10 300 $B "$@" 2>>$OUT/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    r = j.get('roofline', {})
    print('%-28s ms %.3f  TFLOP/s %.2f  chol %.2f  bulk_frac %.3f  launches %.0f  logp_err %s' % ('$name', j['ms_per_step'], j['value'], j.get('cholesky_tflops', 0), r.get('frac', 0), r.get('launches_per_step', 0), j.get('logp_rel_err')))
" | tee -a $OUT/sweep.log
}
# config 4 (N=32768)
for sb in 1 2; do for nb in 1024 512; do
  run "c4 NB=$nb SB=$sb" G3_NB=$nb G3_SB=$sb -- --steps 4 --warmup 1
done; done
run "c4 NB=1024 SB=2 order0" G3_NB=1024 G3_SB=2 G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_order0.so -- --steps 4 --warmup 1
run "c4 NB=1024 SB=1 order0" G3_NB=1024 G3_SB=1 G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_order0.so -- --steps 4 --warmup 1
# config 3 (N=16384, MAT52+COS d=8)
for sb in 1 2 4; do for nb in 1024 512 256; do
  run "c3 NB=$nb SB=$sb" G3_NB=$nb G3_SB=$sb -- --points 16384 --dims 8 --kernel mat52cos --steps 10 --warmup 2
done; done
# config 2 (N=8192)
for sb in 1 2 4; do for nb in 512 256 128; do
  run "c2 NB=$nb SB=$sb" G3_NB=$nb G3_SB=$sb -- --points 8192 --steps 20 --warmup 3
done; done
# smaller sizes
for n in 4096 12288 20480; do for sb in 1 2; do
  run "n$n SB=$sb" G3_SB=$sb -- --points $n --steps 10 --warmup 2
done; done
echo done >> $OUT/sweep.log

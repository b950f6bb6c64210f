#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 $B "$@" 2>>$OUT/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    r = j.get('roofline', {})
    print('%-34s ms %.3f  TFLOP/s %.2f  chol %.2f  bulk_frac %.3f  launches %.0f  logp_err %s' % ('$name', j['ms_per_step'], j['value'], j.get('cholesky_tflops', 0), r.get('frac', 0), r.get('launches_per_step', 0), j.get('logp_rel_err')))
" | tee -a $OUT/sweep.log
}
for b8 in 0 1; do for min in 4096 1024 256; do
  run "c4 BULK8=$b8 BIGMIN=$min" G3_GEMM_BULK8=$b8 G3_GEMM_BIG_MIN=$min -- --steps 4 --warmup 1
done; done
run "c4 BULK8=1 SB=2 NB=512" G3_GEMM_BULK8=1 G3_SB=2 G3_NB=512 -- --steps 4 --warmup 1
run "c4 BULK8=1 SB=2 NB=1024" G3_GEMM_BULK8=1 G3_SB=2 G3_NB=1024 -- --steps 4 --warmup 1
for b8 in 0 1; do for min in 4096 1024 256; do
  run "c3 BULK8=$b8 BIGMIN=$min" G3_GEMM_BULK8=$b8 G3_GEMM_BIG_MIN=$min -- --points 16384 --dims 8 --kernel mat52cos --steps 10 --warmup 2
  run "c2 BULK8=$b8 BIGMIN=$min" G3_GEMM_BULK8=$b8 G3_GEMM_BIG_MIN=$min -- --points 8192 --steps 20 --warmup 3
done; done
run "c3 BULK8=1 SB=2 NB=512" G3_GEMM_BULK8=1 G3_SB=2 G3_NB=512 -- --points 16384 --dims 8 --kernel mat52cos --steps 10 --warmup 2
run "c5 f32 BULK8=0" G3_GEMM_BULK8=0 -- --f32 --points 65536 --dims 16 --queries 4096 --steps 2 --warmup 1
run "c5 f32 BULK8=1" G3_GEMM_BULK8=1 -- --f32 --points 65536 --dims 16 --queries 4096 --steps 2 --warmup 1

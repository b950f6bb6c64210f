#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3tune}; mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 $B "$@" 2>>$OUT/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('%-40s ms %.3f  TFLOP/s %.2f' % ('$name', j['ms_per_step'], j['value']))
" | tee -a $OUT/sweep.log
}
for n in 4096 6144 8192 12288 16384 24576; do
  st=20; [ $n -ge 12288 ] && st=8
  for nb in 128 256 512 1024; do
    [ $n -le 4096 ] && [ $nb -ge 1024 ] && continue
    [ $n -ge 16384 ] && [ $nb -le 128 ] && continue
    for tail in 6 10 16; do
      run "n$n NB=$nb TAIL=$tail" G3_NB=$nb G3_NB_TAIL=$tail -- --points $n --steps $st --warmup 3
    done
  done
done
echo "== batched chains" | tee -a $OUT/sweep.log
timeout -k 10 300 python scripts/chain_bench.py 2>&1 | grep -v amdgpu | tee -a $OUT/sweep.log

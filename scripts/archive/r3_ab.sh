#!/bin/bash
# same-box A/B of library variants (G3_LIB_PATH): usage r3_ab.sh <out> <variant .so name or "-" for the product> ...
OUT=gpurun_out/${1:-r3ab}; shift; mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api --skip-events"
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" == "-" ]; then unset G3_LIB_PATH; else export G3_LIB_PATH=$PWD/g3py_amd/lib/$v; fi
  for cfg in "--points 8192 --steps 30 --warmup 5" "--points 16384 --dims 8 --kernel mat52cos --steps 10 --warmup 2" "--steps 5 --warmup 2"; do
    timeout -k 10 200 $B $cfg 2>>$OUT/err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', j['config']['N'], 'ms %.3f' % j['ms_per_step'], 'logp_err', j.get('logp_rel_err'))" | tee -a $OUT/ab.log
  done
done
done

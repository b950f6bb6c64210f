#!/bin/bash
# same-box A/B of an environment knob: r3_env_ab.sh <out> VAR v1 v2 ...   (configs 2 / 3 / 4, two rounds)
OUT=gpurun_out/${1:-r3env}; VAR=$2; shift 2; mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api --skip-events"
for rep in 1 2; do
for v in "$@"; do
  for cfg in "--points 8192 --steps 30 --warmup 5" "--points 16384 --dims 8 --kernel mat52cos --steps 10 --warmup 2" "--points 24576 --steps 6 --warmup 2" "--steps 6 --warmup 2"; do
    env $VAR=$v timeout -k 10 200 $B $cfg 2>>$OUT/err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$VAR=$v', j['config']['N'], 'ms %.3f' % j['ms_per_step'], 'logp_err', j.get('logp_rel_err'), 'logp %.9f' % j['logp'])" | tee -a $OUT/ab.log
  done
done
done

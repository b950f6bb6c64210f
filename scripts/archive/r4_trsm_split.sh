#!/bin/bash
# A/B of the launch-level split of the tall panel solve (G3_TRSM_SPLIT_MIN rows, G3_TRSM_SPLIT_N widest stripe launch)
cd "$(dirname "$0")/.."
ms() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('%.3f ms (median %.3f)  err %.1e' % (d['ms_per_step'], d.get('ms_per_step_median', float('nan')), d.get('logp_rel_err', float('nan'))))"; }
B="python bench.py --cpu-n 0 --no-api --no-measure-traffic --skip-events"
for cfg in "c4 --steps 6 --warmup 2" "c3 --points 16384 --dims 8 --kernel mat52cos --steps 8 --warmup 2" "n24576 --points 24576 --steps 6 --warmup 2"; do
  set -- $cfg; name=$1; shift
  echo "$name base                    : $($B "$@" 2>/dev/null | ms)"
  for smin in 4096 8192 16384; do
    for sn in 512 256; do
      echo "$name split_min=$smin split_n=$sn : $(G3_TRSM_SPLIT_MIN=$smin G3_TRSM_SPLIT_N=$sn $B "$@" 2>/dev/null | ms)"
    done
  done
  echo "$name base again              : $($B "$@" 2>/dev/null | ms)"
done

#!/bin/bash
# one-GPU configurations with and without the stream-placement probe, same box: scripts/r5_bench_ab.sh <outdir>
OUT=${1:-gpurun_out/r5_ab}; mkdir -p $OUT
B="python bench.py --cpu-n 0 --no-measure-traffic --no-api"
ms() { python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms median, %.3f mean, frac %.3f' % (j['ms_per_step_median'], j['ms_per_step'], j['roofline']['frac']))"; }
for rep in 1 2; do
for pr in 2 0; do
  echo "c2 G3_PROBE=$pr rep $rep: $(G3_PROBE=$pr timeout -k 10 200 $B --points 8192 --steps 30 --warmup 5 2>$OUT/c2_p${pr}_$rep.err | ms)"
  echo "c3 G3_PROBE=$pr rep $rep: $(G3_PROBE=$pr timeout -k 10 200 $B --points 16384 --dims 8 --kernel mat52cos --steps 15 --warmup 3 2>$OUT/c3_p${pr}_$rep.err | ms)"
  echo "c4 G3_PROBE=$pr rep $rep: $(G3_PROBE=$pr timeout -k 10 200 $B --steps 10 --warmup 2 2>$OUT/c4_p${pr}_$rep.err | ms)"
done
done
grep -h "placement" $OUT/*.err | sort | uniq -c

#!/bin/bash
OUT=gpurun_out/${1:-r3f}; mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api"
for nb in 1024 512; do
echo "== native driver, one rank, RCCL nb=$nb" | tee -a $OUT/dist.log
G3_FORCE_DIST=1 timeout -k 10 300 $B --steps 4 --warmup 1 --panel $nb 2>>$OUT/dist.err | tee $OUT/bench_dist1_native_rccl_$nb.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ms', j['ms_per_step'], 'logp_err', j.get('logp_rel_err'), 'frac', j['roofline']['frac'])" | tee -a $OUT/dist.log
done
timeout -k 10 300 python -m pytest tests/test_gpu_distributed.py -x -q -k "native or hip_block" 2>&1 | tail -3

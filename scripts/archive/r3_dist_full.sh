#!/bin/bash
# full-size (N=32768) rehearsal of the native driver with 2 and 4 ranks sharing the GPU over gloo, logp pin + gradient
OUT=gpurun_out/${1:-r3full}; mkdir -p $OUT
for P in 2 4; do
echo "== $P ranks" | tee -a $OUT/log
G3_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 2960$P \
  bench.py --gpus $P --grad --steps 1 --warmup 1 2>>$OUT/err > $OUT/bench_dist${P}.json || { tail -5 $OUT/err; exit 1; }
python - <<PY | tee -a $OUT/log
import json
j=[json.loads(l) for l in open('$OUT/bench_dist${P}.json') if l.startswith('{')][-1]
print('ms', j['ms_per_step'], 'logp_rel_err', j['logp_rel_err'], 'dlogp', {k: v for k, v in j['dlogp'].items() if k.endswith('_ms')}, j['dlogp']['grad_natural'][:3])
PY
done

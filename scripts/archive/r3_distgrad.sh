#!/bin/bash
# dlogp on the distributed covariance: one rank through RCCL at full size against the one-GPU dlogp, then 2 and 4 ranks
# sharing the GPU over gloo (callback transport) at N=16384
OUT=gpurun_out/${1:-r3grad}; mkdir -p $OUT
B="python bench.py --no-measure-traffic --cpu-n 0 --no-api --grad"
echo "== one GPU, in-library dlogp (reference numbers)" | tee -a $OUT/log
timeout -k 10 400 $B --steps 2 --warmup 1 2>>$OUT/err > $OUT/bench_grad_1gpu.json || exit 1
echo "== native driver, one rank, RCCL, --grad" | tee -a $OUT/log
G3_FORCE_DIST=1 timeout -k 10 400 $B --steps 2 --warmup 1 2>>$OUT/err > $OUT/bench_grad_dist1_rccl.json || exit 1
for P in 2 4; do
echo "== native driver, $P ranks on one GPU over gloo, N=16384 --grad" | tee -a $OUT/log
G3_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 2950$P \
  bench.py --gpus $P --no-measure-traffic --cpu-n 0 --no-api --grad --points 16384 --steps 1 --warmup 1 --panel 512 2>>$OUT/err > $OUT/bench_grad_dist${P}_gloo.json || exit 1
done
G3_FORCE_DIST=1 timeout -k 10 400 python bench.py --no-measure-traffic --cpu-n 0 --no-api --grad --points 16384 --steps 2 --warmup 1 --panel 512 2>>$OUT/err > $OUT/bench_grad_dist1_16384.json || exit 1
python - <<PY | tee -a $OUT/log
import json
def last(p):
    return [json.loads(l) for l in open(p) if l.startswith('{')][-1]
a = last('$OUT/bench_grad_1gpu.json'); b = last('$OUT/bench_grad_dist1_rccl.json')
print('one GPU    : step %.1f ms, dlogp %.1f ms' % (a['ms_per_step'], a['dlogp']['ms']))
print('dist 1 rank: step %.1f ms, step in gradient mode %.1f ms, dlogp %.1f ms' % (b['ms_per_step'], b['dlogp']['step_grad_mode_ms'], b['dlogp']['dlogp_ms']))
ga, gb = a['dlogp']['grad_natural'], b['dlogp']['grad_natural']
print('max rel diff of the parameter sums:', max(abs(x - y) / abs(x) for x, y in zip(ga, gb)))
r = last('$OUT/bench_grad_dist1_16384.json')
for P in (2, 4):
    c = last('$OUT/bench_grad_dist%d_gloo.json' % P)
    print('N=16384 %d ranks (gloo): max rel diff vs one rank:' % P, max(abs(x - y) / abs(x) for x, y in zip(r['dlogp']['grad_natural'], c['dlogp']['grad_natural'])),
          'logp', c['logp'], r['logp'])
PY

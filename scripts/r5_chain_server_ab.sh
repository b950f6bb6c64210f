#!/bin/bash
# chain-server variant (scripts/variants/chain_server.inc) against the default sweep, configs 2 and 3, under different stream
# histories of the process (G3_BENCH_HIPRIO, G3_BENCH_IDLE_STREAMS idle streams created first): scripts/r5_chain_server_ab.sh <outdir>
OUT=${1:-gpurun_out/r5_chain}; mkdir -p $OUT
B="python bench.py --cpu-n 0 --no-measure-traffic --no-api"
ms() { python -c "import json,sys; L=[l for l in sys.stdin if l.startswith('{')]; j=json.loads(L[-1]) if L else None; print('%.3f ms median, %.3f mean, logp_rel_err %.1e' % (j['ms_per_step_median'], j['ms_per_step'], j.get('logp_rel_err') or 0) if j else 'NO LINE')"; }
for hp in 1 0; do
for idle in 0 3; do
for lib in chain default; do
  if [ $lib == chain ]; then export G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_chain.so G3_CHAIN=1; else unset G3_LIB_PATH G3_CHAIN; fi
  echo "c2 lib=$lib hiprio=$hp idle=$idle: $(G3_BENCH_HIPRIO=$hp G3_BENCH_IDLE_STREAMS=$idle timeout -k 10 120 $B --points 8192 --steps 30 --warmup 5 2>$OUT/c2_${lib}_${hp}_${idle}.err | ms)"
  echo "c3 lib=$lib hiprio=$hp idle=$idle: $(G3_BENCH_HIPRIO=$hp G3_BENCH_IDLE_STREAMS=$idle timeout -k 10 120 $B --points 16384 --dims 8 --kernel mat52cos --steps 12 --warmup 3 2>$OUT/c3_${lib}_${hp}_${idle}.err | ms)"
done
done
done

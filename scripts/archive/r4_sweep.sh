#!/bin/bash
cd "$(dirname "$0")/.."
export G3_BENCH_HIPRIO=0
W4=$PWD/g3py_amd/lib/libg3hip_w4.so
ms() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('%.3f ms  err %.1e' % (d['ms_per_step'], d.get('logp_rel_err', float('nan'))))"; }
B="python bench.py --cpu-n 0 --no-api --no-measure-traffic --skip-events"
for cfg in "n4096 --points 4096 --steps 30 --warmup 3" "c2 --points 8192 --steps 20 --warmup 3" "n12288 --points 12288 --steps 10 --warmup 2" "c3 --points 16384 --dims 8 --kernel mat52cos --steps 8 --warmup 2" "c4 --steps 4 --warmup 1"; do
  set -- $cfg; name=$1; shift
  echo "$name classic            : $(G3_CHAIN=0 $B "$@" 2>/dev/null | ms)"
  for wgs in 16 32 48 64; do
    echo "$name chain d256 wgs=$wgs : $(G3_CHAIN_WGS=$wgs $B "$@" 2>/dev/null | ms)"
    echo "$name chain d128 wgs=$wgs : $(G3_LIB_PATH=$W4 G3_CHAIN_WGS=$wgs $B "$@" 2>/dev/null | ms)"
  done
done

#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --cpu-n 0 --no-api --no-measure-traffic --points 8192 --steps 20 --warmup 3 --skip-events"
ms() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('%.3f ms' % d['ms_per_step'])"; }
export G3_CHAIN_ALLMASK=1
for hp in 1 0; do
export G3_BENCH_HIPRIO=$hp
echo "allmask hiprio $hp chain d256 wgs 16 : c2 $(G3_CHAIN_WGS=16 $B 2>/dev/null | ms)"
done
G3_CHAIN_WGS=16 timeout -k 10 120 python scripts/r4_chain_check.py 2048 4096 8192

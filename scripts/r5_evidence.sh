#!/bin/bash
# Round-5 evidence on the final tree (one box): bench lines of configs 2 / 3 / 5-shape / gradient, the one-rank native driver,
# compute-side replays P = 2 / 4 / 8 (config 4) and P = 8 (config 5 shape), chain tables.  usage: scripts/r5_evidence.sh [part ...]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05e; mkdir -p $OUT $R/gpurun_out/replay
parts=${@:-bench replay chains}
B="python bench.py --cpu-n 0 --no-measure-traffic"
for p in $parts; do
case $p in
bench)
  timeout -k 10 200 $B --points 8192 --steps 30 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"
  timeout -k 10 200 $B --points 16384 --dims 8 --kernel mat52cos --steps 15 --warmup 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "c3 rc=$?"
  timeout -k 10 400 $B --f32 --points 65536 --dims 16 --queries 4096 --draws 16 --steps 3 --warmup 1 --no-api > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"
  timeout -k 10 300 $B --grad --steps 3 --warmup 1 --no-api > $OUT/bench_grad.json 2> $OUT/bench_grad.err; echo "grad rc=$?"
  ;;
replay)
  R5_VARIANTS=fullinv timeout -k 10 500 python scripts/r5_replay.py c4 2 4 8 > $OUT/replay_c4.txt 2>&1; echo "replay c4 rc=$?"
  R5_VARIANTS=fullinv timeout -k 10 500 python scripts/r5_replay.py c5 8 > $OUT/replay_c5.txt 2>&1; echo "replay c5 rc=$?"
  ;;
chains)
  timeout -k 10 200 python scripts/r4_chain_small.py > $OUT/chain_small.txt 2>&1; echo "chain small rc=$?"
  timeout -k 10 300 python scripts/r5_chain_medium.py > $OUT/chain_medium.txt 2>&1; echo "chain medium rc=$?"
  timeout -k 10 200 python scripts/r5_chain_host.py > $OUT/chain_host.txt 2>&1; echo "chain host rc=$?"
  ;;
esac
done

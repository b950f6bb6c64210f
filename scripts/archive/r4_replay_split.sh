#!/bin/bash
# the 8-rank schedule (replay, config 4, nb = 1024) with the panel solve split at the launch level
cd "$(dirname "$0")/.."
run() { label=$1; shift; echo "== $label"; env "$@" R4_NB=1024 timeout -k 10 200 python scripts/r4_replay.py c4 8 2>&1 | grep -E "P 8"; python - <<'PY'
import json
d=json.load(open('gpurun_out/replay/r04_replay_c4_P8_nb1024.json'))
pr=d['per_rank']
print('   bulk %.1f-%.1f ms, solve %.1f-%.1f ms, rank step %.1f-%.1f ms' % (min(r['bulk_gemm_ms'] for r in pr), max(r['bulk_gemm_ms'] for r in pr), min(r['solve_ms'] for r in pr), max(r['solve_ms'] for r in pr), min(r['ms_per_step'] for r in pr), max(r['ms_per_step'] for r in pr)))
PY
}
run "default"
run "split from 1024 rows, 512 columns" G3_TRSM_SPLIT_MIN=1024 G3_TRSM_SPLIT_N=512
run "split from 1024 rows, 256 columns" G3_TRSM_SPLIT_MIN=1024 G3_TRSM_SPLIT_N=256
run "split from 2048 rows, 256 columns" G3_TRSM_SPLIT_MIN=2048 G3_TRSM_SPLIT_N=256
run "split from 1024 rows, 128 columns" G3_TRSM_SPLIT_MIN=1024 G3_TRSM_SPLIT_N=128

#!/bin/bash
# per-rank efficiency of the 8-rank schedule under tile / stripe knobs (replay transport, config 4, nb = 1024)
cd "$(dirname "$0")/.."
run() { label=$1; shift; echo "== $label"; env "$@" R4_NB=1024 timeout -k 10 200 python scripts/r4_replay.py c4 8 2>&1 | grep -E "P 8|in-library"; }
run "default"
run "big tile from 1024 tiles" G3_GEMM_BIG_MIN=1024
run "big tile from 512 tiles" G3_GEMM_BIG_MIN=512
run "big tile from 256 tiles" G3_GEMM_BIG_MIN=256
run "16-row stripes up to 8192 rows" G3_TRSM_THIN_MAX=8192
run "16-row stripes up to 8192 rows + big tile from 512" G3_TRSM_THIN_MAX=8192 G3_GEMM_BIG_MIN=512
run "no residency cap" G3_SIDE_LDS=0

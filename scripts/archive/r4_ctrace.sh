#!/bin/bash
# chain-server timelines for a few builds / worker counts (development aid)
cd "$(dirname "$0")/.."
for v in ctrace ctrace2; do
  for wgs in 16 32; do
    echo "=== lib $v wgs $wgs" 
    G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_$v.so G3_CHAIN_WGS=$wgs timeout -k 10 120 python scripts/r4_chain_trace.py 8192 2>&1 | sed -n '/--- panel 3/,/--- panel 4/p'
  done
done
echo "=== exclusive workers (160 KB LDS), 32"
G3_LIB_PATH=$PWD/g3py_amd/lib/libg3hip_ctrace.so G3_CHAIN_WGS=32 G3_CHAIN_LDS=163000 timeout -k 10 120 python scripts/r4_chain_trace.py 8192 2>&1 | sed -n '/--- panel 3/,/--- panel 4/p'
for wgs in 16 32 48; do G3_CHAIN_WGS=$wgs timeout -k 10 120 python scripts/r4_chain_check.py 2048 4096 8192; done
G3_CHAIN_WGS=32 G3_CHAIN_LDS=163000 timeout -k 10 120 python scripts/r4_chain_check.py 2048 4096 8192
G3_CHAIN=0 timeout -k 10 120 python scripts/r4_chain_check.py 2048 4096 8192

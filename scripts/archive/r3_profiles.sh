#!/bin/bash
# round-3 profile collection (one gpurun call): traces + stats for configs 4/2/3, PMC passes, bench lines.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --cpu-n 0 --no-api --no-measure-traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B --steps 3 --warmup 1 > $OUT/trace.log 2>&1; echo "trace c4 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $B --points 8192 --steps 6 --warmup 1 > $OUT/c2.log 2>&1; echo "trace c2 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $B --points 16384 --dims 8 --kernel mat52cos --steps 4 --warmup 1 > $OUT/c3.log 2>&1; echo "trace c3 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 1 --warmup 0 --skip-events > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 1 --warmup 0 --skip-events > $OUT/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $B --steps 1 --warmup 0 --skip-events > $OUT/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
cd $R
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
timeout -k 10 300 python bench.py --points 8192 --steps 30 --warmup 5 --cpu-n 8192 --no-measure-traffic > $OUT/bench_c2.json 2>> $OUT/bench.err; echo "c2 rc=$?"
timeout -k 10 300 python bench.py --points 16384 --dims 8 --kernel mat52cos --steps 10 --warmup 2 --cpu-n 0 --no-measure-traffic > $OUT/bench_c3.json 2>> $OUT/bench.err; echo "c3 rc=$?"
timeout -k 10 400 python bench.py --f32 --points 65536 --dims 16 --queries 4096 --steps 3 --warmup 1 --cpu-n 0 > $OUT/bench_c5.json 2>> $OUT/bench.err; echo "c5 rc=$?"
timeout -k 10 300 python bench.py --grad --steps 3 --warmup 1 --cpu-n 0 --no-api --no-measure-traffic > $OUT/bench_grad.json 2>> $OUT/bench.err; echo "grad rc=$?"
G3_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-n 0 --no-api --no-measure-traffic > $OUT/bench_dist1_nccl.json 2>> $OUT/bench.err; echo "dist1 native rc=$?"
G3_FORCE_DIST=1 G3_DIST_DRIVER=python G3_DIST_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-n 0 --no-api --no-measure-traffic > $OUT/bench_dist1_python_nccl.json 2>> $OUT/bench.err; echo "dist1 python rc=$?"
tail -3 $OUT/bench.err

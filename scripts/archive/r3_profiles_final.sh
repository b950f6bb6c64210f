#!/bin/bash
# final-tree refresh of the headline profile: kernel trace + stats, PMC passes (fetch / write / MFMA busy)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03}; rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --cpu-n 0 --no-api --no-measure-traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B --steps 3 --warmup 1 > $OUT/trace.log 2>&1; echo "trace c4 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 1 --warmup 0 --skip-events > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 1 --warmup 0 --skip-events > $OUT/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $B --steps 1 --warmup 0 --skip-events > $OUT/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
grep -h '^{' $OUT/trace.log | cut -c1-200

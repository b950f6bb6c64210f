#!/bin/bash
# round-3 counter passes on the stand-alone bulk update (30720^2 x K lower, fp64): what the MFMA pipe waits for.
# One small counter group per pass (rocprofv3 --pmc, kernel-trace only -- the guide's rule).
OUT=$PWD/gpurun_out/${1:-r3pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
K=${2:-1024}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/scripts/gemm_bench.py 30720 30720 $K 1 > $OUT/$tag.log 2>&1
  echo "$tag rc=$?" >> $OUT/pmc.log
done
python3 - <<PY >> $OUT/pmc.log
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt_kernel' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print('%-32s per launch %.4e  (%d launches)' % (k, tot[k] / n[k], n[k]))
PY
cat $OUT/pmc.log

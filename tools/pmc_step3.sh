#!/bin/bash
# SQ / LDS counters of d2q9_step3 (LDS windows) on 8192x8192 and 1024x1024
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for sz in "8192 8192 60" "1024 1024 3000"; do
  set -- $sz
  TAG=s3_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_${TAG}_stats -o s -- python3 $REPO/tools/run_case.py $1 $2 $3 > $OUT/pmc_${TAG}_stats.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_${TAG}_sq -o c -- python3 $REPO/tools/run_case.py $1 $2 $3 > $OUT/pmc_${TAG}_sq.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_${TAG}_lds -o c -- python3 $REPO/tools/run_case.py $1 $2 $3 > $OUT/pmc_${TAG}_lds.log 2>&1 || echo "lds pass failed"
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_${TAG}_grbm -o c -- python3 $REPO/tools/run_case.py $1 $2 $3 > $OUT/pmc_${TAG}_grbm.log 2>&1 || echo "grbm pass failed"
  echo "== $1 x $2"; python3 $REPO/tools/pmc_report.py $TAG d2q9_step3
done

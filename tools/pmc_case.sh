#!/bin/bash
# tools/pmc_case.sh <tag> <run_case.py args...> : SQ issue/wait counters + FETCH/WRITE sizes for one configuration
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_${TAG}_stats -o s -- python3 $REPO/tools/run_case.py "$@" > $OUT/pmc_${TAG}_stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_${TAG}_sq -o c -- python3 $REPO/tools/run_case.py "$@" > $OUT/pmc_${TAG}_sq.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch -o c -- python3 $REPO/tools/run_case.py "$@" > $OUT/pmc_${TAG}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write -o c -- python3 $REPO/tools/run_case.py "$@" > $OUT/pmc_${TAG}_write.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_${TAG}_tcc -o c -- python3 $REPO/tools/run_case.py "$@" > $OUT/pmc_${TAG}_tcc.log 2>&1 || echo "tcc pass failed"

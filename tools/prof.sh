#!/bin/bash
# tools/prof.sh <tag> <python script and args ...> — rocprofv3 evidence for one command, run on the GPU box (gpurun):
#   pass 1  --kernel-trace --stats                 per-kernel average duration
#   pass 2  --pmc FETCH_SIZE                       HBM-side read traffic   (separate passes: FETCH_SIZE and WRITE_SIZE do
#   pass 3  --pmc WRITE_SIZE                       HBM-side write traffic   not fit the TCC counter slots together)
#   pass 4  --pmc SQ_* (8 counters)                VALU issue share, wait shares
#   pass 5  --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum    L2 hit rate
#   pass 6  --pmc FETCH_SIZE on a 1 GiB copy       calibration of the x2 on gfx950 (MI355X_MICROARCH.md, HBM section)
# The program itself follows `--` (python3 <script>): never env/bash -c under rocprofv3 on this pool.
# tools/prof_report.py <tag> <kernel substring> <nx> <ny> condenses the passes into profiles/<tag>.txt (+ traffic.json).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args..., then -- is added here
  local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -o p -- python3 "${CMD[@]}" > $OUT/prof_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/prof_${TAG}_$name.log; return 1; }
  echo "pass $name ok: $(tail -1 $OUT/prof_${TAG}_$name.log | cut -c1-160)"
}
CMD=("$@")
run stats --kernel-trace --stats || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum || echo "tcc pass failed (optional)"
CMD=($REPO/tools/calib_copy.py)
run calib --pmc FETCH_SIZE || echo "calibration pass failed (optional)"

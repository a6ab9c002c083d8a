#!/bin/bash
# rocprofv3 evidence for bench.py's numbers (run on the GPU box through gpurun).
#   pass 1: --kernel-trace --stats   -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE         -> HBM read traffic   (separate passes: FETCH_SIZE and WRITE_SIZE
#   pass 3: --pmc WRITE_SIZE         -> HBM write traffic   do not fit the TCC counter slots together)
# Outputs land under gpurun_out/prof_*/ ; tools/summarize_profile.py condenses them into profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=${1:-r01}
EXTRA=${2:-}
ARGS="--no-cpu-baseline --no-extra $EXTRA"  # default steps/warmup: the very command whose JSON line is reported
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -o stats -- python3 $REPO/bench.py $ARGS > $OUT/prof_${TAG}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -o fetch -- python3 $REPO/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra $EXTRA > $OUT/prof_${TAG}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -o write -- python3 $REPO/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra $EXTRA > $OUT/prof_${TAG}_write.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_calib -o calib -- python3 $REPO/tools/calib_copy.py > $OUT/prof_${TAG}_calib.log 2>&1 || exit 1
ls -R $OUT/prof_${TAG}_* | head -40

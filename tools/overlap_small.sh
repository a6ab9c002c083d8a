#!/bin/bash
# kernel trace of small ring-of-one slabs (d2q9_multi, halo depth 8): where a launch set's time goes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for sz in "1024 512" "1024 128"; do
  set -- $sz
  rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap_s$2 -o ring -- python3 $REPO/tools/run_ring.py $1 $2 1600 ${TRANSPORT:-peer} > $OUT/overlap_s$2.log 2>&1 || exit 1
  echo "== $1 x $2"; tail -1 $OUT/overlap_s$2.log
  python3 $REPO/tools/overlap_report.py $OUT/overlap_s$2/ring_kernel_trace.csv
done

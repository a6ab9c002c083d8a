#!/bin/bash
# kernel trace of a ring-of-one slab (edge launch -> halo exchange (TRANSPORT=peer|rccl|copy) overlapping the interior launch)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap -o ring -- python3 $REPO/tools/run_ring.py 8192 1024 200 ${TRANSPORT:-peer} > $OUT/overlap.log 2>&1 || exit 1
python3 $REPO/tools/overlap_report.py $OUT/overlap/ring_kernel_trace.csv
tail -1 $OUT/overlap.log

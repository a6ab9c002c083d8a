#!/bin/bash
# ad-hoc: slab mode (ring of one over RCCL): per-rank rate and kernel-trace overlap report
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for sz in "8192 1024 480" "8192 2048 480" "8192 4096 240" "4096 1024 960"; do python3 $REPO/tools/run_ring.py $sz 2>&1 | grep "ring of one" || exit 1; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap_rsv -o ring -- python3 $REPO/tools/run_ring.py 8192 1024 200 > $OUT/overlap_rsv.log 2>&1 || exit 1
python3 $REPO/tools/overlap_report.py $OUT/overlap_rsv/ring_kernel_trace.csv

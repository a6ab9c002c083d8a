#!/bin/bash
# ad-hoc: rows per edge chunk (LBM_EDGE_CHUNK) in slab mode: ring-of-one rate and kernel-trace overlap report
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ec in 3 1; do
  export LBM_EDGE_CHUNK=$ec
  echo "== LBM_EDGE_CHUNK=$ec"
  for sz in "8192 1024 480" "8192 2048 480" "4096 512 960" "2048 256 1920"; do python3 $REPO/tools/run_ring.py $sz || exit 1; done
  rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap_ec$ec -o ring -- python3 $REPO/tools/run_ring.py 8192 1024 200 > $OUT/overlap_ec$ec.log 2>&1 || exit 1
  python3 $REPO/tools/overlap_report.py $OUT/overlap_ec$ec/ring_kernel_trace.csv
done

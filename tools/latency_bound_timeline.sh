#!/bin/bash
# tools/latency_bound_timeline.sh — kernel timelines of the latency-bound sizes: how long a launch of the
# kernel that runs there takes against the launch period (= what a persistent kernel could at most take away), next to
# tools/barrier_probe (what its grid barrier would cost).  Run on the GPU box; writes gpurun_out/r03/latency_*.txt.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$REPO/tools/barrier_probe > $OUT/latency_barrier_probe.txt 2>&1 || echo "barrier probe failed"
cat $OUT/latency_barrier_probe.txt
for case in "128 128 8000" "256 256 8000" "1024 1024 4000" "1024 128 8000 force_halo=1 transport=3"; do
  tag=$(echo $case | tr ' =' '__')
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -o t -- python3 $REPO/tools/run_case.py $case > $OUT/trace_$tag.log 2>&1 || { echo "trace $case failed"; tail -3 $OUT/trace_$tag.log; continue; }
  echo "== $case: $(tail -1 $OUT/trace_$tag.log)"
  python3 $REPO/tools/timeline.py $OUT/trace_$tag/t_kernel_trace.csv | head -12
done > $OUT/latency_timelines.txt 2>&1
cat $OUT/latency_timelines.txt

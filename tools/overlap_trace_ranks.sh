#!/bin/bash
# tools/overlap_trace_ranks.sh N [nx rows steps transport] — halo/compute overlap trace of an N-rank run (BASELINE config 5:
# "8xMI355X weak-scaling MLUPS + halo/compute overlap trace"): every rank is its own `rocprofv3 --kernel-trace -- python3
# tools/run_ring_rank.py ...` (the profiled program follows `--` directly: no env / bash -c / launcher hop), started here in
# the background; tools/overlap_report.py then reads each rank's kernel trace: launch-set period, edge / exchange kernels
# and how much of them runs under an interior kernel.  On an N-GPU node: transport rccl (or peer) with one rank per
# device.  On a one-GPU box N ranks share the device (transport peer: HIP IPC between real processes; at most 6).
N=${1:-2}; NX=${2:-8192}; ROWS=${3:-1024}; STEPS=${4:-200}; TR=${5:-peer}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/overlap_ranks
mkdir -p $OUT
PORT=$((20000 + RANDOM % 20000))
cd /tmp && export TMPDIR=/tmp
pids=()
for ((r = 0; r < N; r++)); do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/rank$r -o ring -- python3 $REPO/tools/run_ring_rank.py $r $N $PORT $NX $ROWS $STEPS $TR > $OUT/rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
for ((r = 0; r < N; r++)); do
  echo "== rank $r: $(grep '^rank ' $OUT/rank$r.log | tail -1)"
  [ -f $OUT/rank$r/ring_kernel_trace.csv ] && python3 $REPO/tools/overlap_report.py $OUT/rank$r/ring_kernel_trace.csv || { echo "no trace for rank $r"; tail -5 $OUT/rank$r.log; }
done
exit $rc

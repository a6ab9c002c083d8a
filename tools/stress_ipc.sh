#!/bin/bash
# tools/stress_ipc.sh ["world nx ny nsteps fuse multistep sync" ...] — repeated runs of the multi-process peer ring
# (tests/_ipc_ring.py: WORLD processes on the one GPU, HIP IPC, peer stores), each bounded; prints one line per run
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r02
i=0
if [ $# -eq 0 ]; then set -- "4 1024 256 203 0 8 2" "4 1024 256 203 0 8 0" "3 1024 384 163 0 8 2" "2 1024 256 99 0 8 2" "4 1024 512 83 0 8 1"; fi
for cfg in "$@"; do
  set -- $cfg
  i=$((i+1))
  port=$((29600 + i))
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tests/_ipc_ring.py $2 $3 $4 $5 $6 $7 > gpurun_out/r02/stress_$i.log 2>&1
  rc=$?
  echo "run $i cfg [$cfg] rc=$rc: $(grep -c 'ipc-ring ok' gpurun_out/r02/stress_$i.log) ok $(grep -v Gloo gpurun_out/r02/stress_$i.log | grep -m1 'AssertionError\|LBMError' | cut -c1-300)"
done

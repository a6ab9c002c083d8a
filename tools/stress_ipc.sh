#!/bin/bash
# repeated runs of the multi-process peer ring (in-kernel wait), each bounded
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
i=0
for cfg in "4 1024 256 203 0 8 2" "4 1024 256 203 0 8 2" "4 1024 256 203 0 8 2" "4 1024 256 203 0 8 0" "3 1024 384 163 0 8 2" "4 1024 256 203 0 8 2" "2 1024 256 99 0 8 2" "4 1024 512 83 0 8 2" "4 1024 256 203 0 8 1" "4 1024 256 203 0 8 2"; do
  set -- $cfg
  i=$((i+1))
  port=$((29600 + i))
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tests/_ipc_ring.py $2 $3 $4 $5 $6 $7 > gpurun_out/r02/stress_$i.log 2>&1
  rc=$?
  echo "run $i cfg [$cfg] rc=$rc: $(grep -c 'ipc-ring ok' gpurun_out/r02/stress_$i.log) ok"
  if [ $rc -ne 0 ]; then grep -v Gloo gpurun_out/r02/stress_$i.log | grep -i "error\|assert\|differ\|Traceback\|LBM" | head -20; fi
done

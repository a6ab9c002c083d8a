#!/bin/bash
# tools/campaign.sh <tag> — the measurement set a round's documents quote, in one gpurun call (about 10 minutes):
# bench lines (400 steps, driver style), the profiler passes of the headline grid and of the 1024x1024 input, the per-size
# table, the scaling projection from ring-of-one runs (peer and RCCL), the reference's acceptance procedure.
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_400.json 2> $OUT/bench_400.err; echo "bench 400: $(cut -c1-120 $OUT/bench_400.json)"
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver: $(cut -c1-120 $OUT/bench_driver.json)"
bash tools/prof.sh ${TAG}b $REPO/bench.py --no-cpu-baseline --no-extra --no-cold 2>&1 | tail -6
bash tools/prof.sh ${TAG}c3 $REPO/bench.py --nx 1024 --ny 1024 --workload tiled --steps 2560 --warmup 256 --no-cpu-baseline --no-extra --no-cold 2>&1 | tail -6   # (d2q9_resident: ten launches of 256 steps)
cd $REPO
python tools/ab.py --sizes 128x128,256x256,512x512,768x768,1024x1024,1536x1024,2048x2048,3072x2048,4096x4096,6144x6144,8192x8192,16384x8192 > $OUT/sizes.txt 2>&1; cat $OUT/sizes.txt
python tools/ab.py --sizes 8192x8192 --opts "fuse=4;fuse=3;fuse=1;fuse=0" >> $OUT/sizes.txt 2>&1; tail -4 $OUT/sizes.txt
python tools/scaling_projection.py peer > $OUT/proj_peer.txt 2>&1; cat $OUT/proj_peer.txt
python tools/scaling_projection.py rccl > $OUT/proj_rccl.txt 2>&1; cat $OUT/proj_rccl.txt
python tools/acceptance.py > $OUT/acceptance.txt 2>&1; tail -12 $OUT/acceptance.txt

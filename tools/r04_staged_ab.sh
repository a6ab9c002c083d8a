#!/bin/bash
# round 4: RCCL launch sets of the deep window kernel, staged (one launch per set) against two streams, ring of one
set -e
python tools/ab.py --sizes 8192x1024,8192x2048,8192x4096 --ring rccl,peer --opts ";compact=0"

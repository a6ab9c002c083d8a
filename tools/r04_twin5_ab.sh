#!/bin/bash
# round 4: row slabs with five halo rows exchanged (d2q9_deep_twin<5, ..., PUSH>) against the previous choice per size, ring of one
set -e
python tools/ab.py --sizes 1024x320,1024x400,1024x512,2048x256 --ring peer --opts ";halo_sync=2;fuse=-1,multistep=8;multistep=8,halo_sync=2"
python tools/ab.py --sizes 1024x256,1024x512,2048x1024 --ring rccl --opts ""

#!/bin/bash
# round 4: d2q9_resident (auto from 200K cells where the grid decomposes) against the library's other choice on the latency-bound sizes
set -e
python tools/ab.py --sizes 256x256,128x2048,512x512,1024x512,512x1024,1024x1024,512x2048 --opts "resident=0;resident=1" --workload cavity
python tools/ab.py --sizes 1024x1024 --opts "resident=0;resident=1" --workload tiled

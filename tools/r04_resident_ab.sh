#!/bin/bash
# round 4: d2q9_resident against the library's choice on the latency-bound sizes
set -e
python tools/ab.py --sizes 512x512,1024x1024 --opts ";resident=1" --workload cavity
python tools/ab.py --sizes 512x512,1024x1024 --opts "resident=1" --workload empty

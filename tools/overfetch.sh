#!/bin/bash
# tools/overfetch.sh — read over-fetch of the four-step kernel by strip width: FETCH_SIZE per launch
# and GLUPS for lanes_out = 60 (default), 56 (7 whole 128-B lines written per strip row), 52, 48 on 8192x8192
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lo in 60 56 52 48 62; do
  python3 $REPO/tools/run_case.py 8192 8192 96 lanes_out=$lo | tail -1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/of_$lo -o f -- python3 $REPO/tools/run_case.py 8192 8192 24 lanes_out=$lo > $OUT/of_$lo.log 2>&1 || { echo "pmc pass failed for $lo"; continue; }
  python3 - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("$OUT/of_$lo/*counter_collection.csv") for r in csv.DictReader(open(f)) if "d2q9_step4" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
m=sum(v)/len(v)
print("lanes_out $lo: FETCH_SIZE %.6g KiB x2 = %.4g bytes per launch = %.3f x the grid (%d launches)" % (m, 2*m*1024, 2*m*1024/(36.0*8192*8192), len(v)))
PY
done

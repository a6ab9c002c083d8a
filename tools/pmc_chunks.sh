#!/bin/bash
# FETCH_SIZE of d2q9_step3 on 8192x8192 against the chunk length: which part of the read over-fetch is the
# chunk-boundary rows (shrinks with longer chunks) and which the strips' edge lines (does not)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ch in 8 16 32 64 128; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_ch$ch -o c -- python3 $REPO/tools/run_case.py 8192 8192 24 chunk_min=$ch chunk_rows=$ch > $OUT/pmc_ch$ch.log 2>&1 || exit 1
  python3 - <<PY
import csv
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$OUT/pmc_ch$ch/c_counter_collection.csv")) if "d2q9_step3" in r["Kernel_Name"]]
ideal=(9*4+1)*8192*8192/1024.0/2   # KiB counted (FETCH_SIZE counts half) for 9 planes + mask
print("chunk %3d rows: FETCH_SIZE %.4g KiB -> reads %.3f GB = %.3f x ideal   %s" % ($ch, sum(v)/len(v), 2*sum(v)/len(v)*1024/1e9, sum(v)/len(v)/ideal, open("$OUT/pmc_ch$ch.log").read().strip().splitlines()[-1]))
PY
done

#!/bin/bash
# kernel trace of the 128x128 input: duration of a d2q9_multi launch against the launch period
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace128 -o t -- python3 $REPO/tools/run_case.py 128 128 8000 > $OUT/trace128.log 2>&1 || exit 1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/trace128/t_kernel_trace.csv")) if "d2q9_multi" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
p=[(int(b["Start_Timestamp"])-int(a["Start_Timestamp"]))/1e3 for a,b in zip(rows[100:-1],rows[101:])]
print("d2q9_multi launches %d  mean duration %.2f us  mean period %.2f us  gap %.2f us" % (len(d), sum(d)/len(d), sum(p)/len(p), sum(p)/len(p)-sum(d[100:])/len(d[100:])))
PY
tail -1 $OUT/trace128.log

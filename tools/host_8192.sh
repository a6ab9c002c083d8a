#!/bin/bash
# the C host (reference CLI) on a synthetic 8192x8192 cavity, 2000 steps, output files suppressed (5.8 GB of text otherwise)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
T=$(mktemp -d)
python3 - "$T" <<'PY'
import sys
d = sys.argv[1]
n = 8192
open(d + "/p.params", "w").write("%d\n%d\n2000\n10\n0.1\n0.005\n1.85\n" % (n, n))
with open(d + "/o.dat", "w") as f:
    for i in range(n):
        f.write("%d 0 1\n%d %d 1\n0 %d 1\n%d %d 1\n" % (i, i, n - 1, i, n - 1, i))
PY
cd $T && LBM_NO_OUTPUT=1 $REPO/d2q9-bgk p.params o.dat
rm -rf $T

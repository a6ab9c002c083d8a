#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [skip] — per-kernel durations and the launch timeline of a rocprofv3
--kernel-trace: for every kernel name count / mean / min duration, then the mean start-to-start period of the most
frequent kernel and `skip`-th..(skip+16)-th dispatches as (start, end) in microseconds relative to the first of them —
where a launch set's time goes (kernel, gap, kernel, ...)."""
import csv
import signal
import sys
from collections import defaultdict

signal.signal(signal.SIGPIPE, signal.SIG_DFL)   # `| head` is fine
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("lbm::", "")[:44]
by = defaultdict(list)
for r in rows:
    by[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, iv in sorted(by.items(), key=lambda kv: -len(kv[1])):
    d = [e - s for s, e in iv]
    per = (iv[-1][0] - iv[len(iv) // 4][0]) / max(1, len(iv) - 1 - len(iv) // 4) if len(iv) > 8 else 0
    print("%-44s n %6d  mean %8.2f us  min %8.2f us  period %8.2f us" % (name, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, per / 1e3))
t0 = int(rows[skip]["Start_Timestamp"])
print("timeline from dispatch %d:" % skip)
for r in rows[skip:skip + 16]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("  %9.2f .. %9.2f  (%7.2f us)  %s  grid %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))

#!/usr/bin/env python3
"""tools/pmc_report.py <tag> [kernel substring] — per-kernel means of the counters collected by tools/pmc_case.sh"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "d2q9_step"
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*" % tag, "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_stats" % tag, "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print("kernel %s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for k in sorted(vals):
    v = vals[k]
    print("%-24s n=%-3d mean=%.6g" % (k, len(v), sum(v) / len(v)))

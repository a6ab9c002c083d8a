#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile.sh (gpurun_out/prof_<tag>_*) into profiles/:
  profiles/<tag>_kernel_stats.csv   the --kernel-trace --stats table, verbatim
  profiles/<tag>_pmc_summary.txt    FETCH_SIZE / WRITE_SIZE per kernel and the derived HBM bytes per launch
  profiles/traffic.json             HBM bytes per launch of the step kernel per workload (read by bench.py)
HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in separate
passes, both are in KiB, and on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes, so
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
The factor 2 is checked on this box with a float4 copy kernel of known size (the `calib` pass)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "8192x8192"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def counters(path):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


shutil.copy(os.path.join(G, "prof_%s_stats" % tag, "stats_kernel_stats.csv"), os.path.join(P, "%s_kernel_stats.csv" % tag))
lines = []
fetch = counters(os.path.join(G, "prof_%s_fetch" % tag, "fetch_counter_collection.csv"))
write = counters(os.path.join(G, "prof_%s_write" % tag, "write_counter_collection.csv"))
step_fetch = step_write = None
for (k, c), v in sorted(fetch.items()):
    lines.append("FETCH_SIZE [KiB]  n=%-4d mean=%-14.6g min=%-14.6g max=%-14.6g %s" % (len(v), sum(v) / len(v), min(v), max(v), k[:90]))
    if "d2q9_step" in k and (step_fetch is None or sum(v) / len(v) > step_fetch):
        step_fetch, step_kernel = sum(v) / len(v), k
for (k, c), v in sorted(write.items()):
    lines.append("WRITE_SIZE [KiB]  n=%-4d mean=%-14.6g min=%-14.6g max=%-14.6g %s" % (len(v), sum(v) / len(v), min(v), max(v), k[:90]))
    if "d2q9_step" in k and (step_write is None or sum(v) / len(v) > step_write):
        step_write = sum(v) / len(v)
calib_path = os.path.join(G, "prof_%s_calib" % tag, "calib_counter_collection.csv")
factor = 2.0
if os.path.exists(calib_path):
    cal = counters(calib_path)
    for (k, c), v in cal.items():
        if "copy_f4" in k:
            kib = sum(v) / len(v)
            lines.append("calibration: copy_f4 of 1 GiB reads FETCH_SIZE = %.6g KiB -> true bytes / counted bytes = %.4f" % (kib, (1 << 30) / (kib * 1024)))
hbm = (factor * step_fetch + step_write) * 1024
with open(os.path.join(G, "prof_%s_stats" % tag, "stats_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        if r["Name"] == step_kernel:
            avg_ns = float(r["AverageNs"])
            lines.append("kernel-trace: %s calls=%s average=%.1f us" % (r["Name"], r["Calls"], avg_ns / 1e3))
nx, ny = (int(v) for v in workload.split("x"))
per_launch = 4 if "step4" in step_kernel else (3 if "step3" in step_kernel else (2 if "step2" in step_kernel else 1))
fused = per_launch > 1
alg = 72.0 * nx * ny * per_launch
lines += ["", "dominant kernel: %s (%d timestep(s) per launch)" % (step_kernel, per_launch), "step kernel, %s: FETCH_SIZE %.6g KiB (x2 on gfx950), WRITE_SIZE %.6g KiB" % (workload, step_fetch, step_write),
          "HBM bytes per launch = (2*FETCH + WRITE)*1024 = %.6g  (reads %.6g, writes %.6g)" % (hbm, 2 * step_fetch * 1024, step_write * 1024),
          "algorithmic bytes per launch = 72 B x %d cells x %d step(s) = %.6g ; traffic / algorithmic = %.4f" % (nx * ny, per_launch, alg, hbm / alg),
          "traffic rate = %.1f GB/s" % (hbm / avg_ns),
          "achieved (algorithmic bytes / average kernel time) = %.1f GB/s" % (alg / avg_ns)]
open(os.path.join(P, "%s_pmc_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
tj_path = os.path.join(P, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj[workload + "/step%d" % per_launch] = {"hbm_bytes_per_launch": round(hbm), "fetch_size_kib": step_fetch, "write_size_kib": step_write,
                "source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)" % tag}
json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)

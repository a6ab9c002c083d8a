#!/usr/bin/env python3
"""tools/prof_report.py <tag> <kernel substring> <nx> <ny> [steps_per_launch] — condenses the rocprofv3 passes of
tools/prof.sh (gpurun_out/prof_<tag>_*) into profiles/<tag>.txt and, for the benchmark grid, profiles/traffic.json
(read by bench.py: roofline.traffic / bound_evidence, marked there as not measured in the run).

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from separate passes, both in
KiB, and on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes: hbm_bytes = (2 FETCH + WRITE) x 1024; the
factor 2 is checked in the same session with a float4 copy of 1 GiB (pass `calib`).  These are the L2's memory-side
request counters: Infinity-Cache hits are included, so for a working set inside the 256 MiB Infinity Cache they say
"bytes that left L2", not "bytes that came from HBM" (the report says which case applies)."""
import collections
import csv
import datetime
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pat, nx, ny = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def counters(name):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(G, "prof_%s_%s" % (tag, name), "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            vals[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return vals


def mean(vals, kernel, counter):
    v = [x for (k, c), xs in vals.items() if kernel in k and c == counter for x in xs]
    return sum(v) / len(v) if v else None


lines = ["rocprofv3 summary `%s` (%s, tools/prof.sh + tools/prof_report.py), kernel filter '%s', grid %dx%d" % (tag, datetime.date.today(), pat, nx, ny), ""]
stats = {}
for f in glob.glob(os.path.join(G, "prof_%s_stats" % tag, "*kernel_stats.csv")):
    lines.append("--kernel-trace --stats (all kernels of the run):")
    for r in csv.DictReader(open(f)):
        lines.append("  %-86s calls %6s  avg %10.2f us  total %6.2f %%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        if pat in r["Name"] and (not stats or float(r["TotalDurationNs"]) > stats["total"]):
            stats = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total": float(r["TotalDurationNs"])}
if not stats:
    sys.exit("no kernel matches %s" % pat)
kern = stats["name"]
per_launch = int(sys.argv[5]) if len(sys.argv) > 5 else (4 if "step4" in kern else 3 if "step3" in kern else 2 if "step2" in kern else 8 if ("multi" in kern or "deep" in kern) else 1)
fetch, write, sq, tcc, cal = (counters(n) for n in ("fetch", "write", "sq", "tcc", "calib"))
fk, wk = mean(fetch, kern, "FETCH_SIZE"), mean(write, kern, "WRITE_SIZE")
cells = nx * ny
lines += ["", "dominant kernel: %s" % kern, "  calls %d, average %.2f us (under the profiler), %d timestep(s) per launch" % (stats["calls"], stats["avg_ns"] / 1e3, per_launch)]
ev = {}
if fk is not None and wk is not None:
    hbm = (2 * fk + wk) * 1024
    model = 73.0 * cells
    alg = 72.0 * cells * per_launch
    fits = 2 * 36.0 * cells < 256 * 2**20
    lines += ["  FETCH_SIZE %.6g KiB (x2 on gfx950) + WRITE_SIZE %.6g KiB -> (2 F + W) x 1024 = %.6g bytes per launch" % (fk, wk, hbm),
              "    reads %.6g (%.3f x the grid), writes %.6g (%.3f x the grid)" % (2 * fk * 1024, 2 * fk * 1024 / (36.0 * cells), wk * 1024, wk * 1024 / (36.0 * cells)),
              "    model (72 + 1 B per cell and launch) %.6g -> traffic / model = %.3f ; algorithmic (72 B x %d steps) %.6g -> traffic / algorithmic = %.3f" % (model, hbm / model, per_launch, alg, hbm / alg),
              "    rate at the profiled duration: %.0f GB/s = %.3f of the 8 TB/s peak%s" % (hbm / stats["avg_ns"], hbm / stats["avg_ns"] / 8000.0,
                  "  (both grids fit the 256 MiB Infinity Cache: bytes that left L2, served mostly on-die, NOT HBM bandwidth)" if fits else "")]
    ev["traffic_bytes"] = round(hbm)
    ev["hbm_util_of_peak_profiled"] = round(hbm / stats["avg_ns"] / 8000.0, 3)
for (k, c), v in cal.items():
    if "copy_f4" in k and c == "FETCH_SIZE":
        kib = sum(v) / len(v)
        lines.append("  calibration: copy of 1 GiB counts FETCH_SIZE = %.6g KiB -> true / counted = %.4f" % (kib, (1 << 30) / (kib * 1024)))
        break
if mean(sq, kern, "SQ_WAVE_CYCLES"):
    g = lambda c: mean(sq, kern, c) or 0.0
    busy = g("SQ_BUSY_CYCLES")
    # SQ_ACTIVE_INST_VALU counts quad-cycles a wave spends issuing VALU; 1024 SIMDs share SQ_BUSY_CYCLES x (SEs)
    lines += ["  SQ (per launch): waves %.0f, VALU wave-instructions %.4g (x 64 lanes = %.1f lane-instructions per cell-step), wave-cycles %.4g" % (
                  g("SQ_WAVES"), g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU") * 64.0 / (cells * per_launch), g("SQ_WAVE_CYCLES")),
              "    of a resident wave's time: issuing %.0f %% (VALU %.0f %%), waiting on memory/LDS/barrier (s_waitcnt) %.0f %%, waiting for an issue slot or a dependency %.0f %%" % (
                  100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
                  100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"))]
    # VALU issue share of the chip: 4 quad-cycles per VALU wave-instruction, 1024 SIMDs, duration in shader cycles from SQ_BUSY_CYCLES / 32 SEs
    if busy:
        dur_cycles = busy / 32.0   # SQ_BUSY_CYCLES is summed over the 32 shader engines (r01: 5.87e7 for a 978-us launch at ~2.0 GHz)
        share = g("SQ_INSTS_VALU") * 4.0 / (1024.0 * dur_cycles) if dur_cycles else 0.0
        lines.append("    VALU issue slots used: %.0f %% (VALU wave-instructions x 4 cycles / (1024 SIMDs x %.3g cycles of the launch))" % (100 * share, dur_cycles))
        ev["valu_issue_share"] = round(share, 3)
        ev["valu_lane_instr_per_cell_step"] = round(g("SQ_INSTS_VALU") * 64.0 / (cells * per_launch), 1)
    ev["wave_time_issuing"] = round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
    ev["wave_time_waitcnt"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3)
if mean(tcc, kern, "TCC_HIT_sum") is not None:
    h, m = mean(tcc, kern, "TCC_HIT_sum"), mean(tcc, kern, "TCC_MISS_sum")
    lines.append("  L2: hits %.4g, misses %.4g -> hit rate %.1f %%; memory-side requests: read %.4g, write %.4g" % (
        h, m, 100 * h / max(1.0, h + m), mean(tcc, kern, "TCC_EA0_RDREQ_sum") or 0, mean(tcc, kern, "TCC_EA0_WRREQ_sum") or 0))
    ev["l2_hit_rate"] = round(h / max(1.0, h + m), 3)
open(os.path.join(P, "%s.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if fk is not None and "traffic_bytes" in ev and len(sys.argv) > 6 and sys.argv[6] == "traffic":
    tj_path = os.path.join(P, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    # the build these counters belong to: the `library` field of the profiled run's own bench line (lbm_version(): a digest
    # of the sources); bench.py attaches the entry only to runs of exactly that build
    version = None
    for ln in open(os.path.join(G, "prof_%s_stats.log" % tag)):
        if ln.startswith('{"metric"'):
            version = json.loads(ln).get("library")
    tj["%dx%d/%s" % (nx, ny, "deep_twin" if "deep_twin" in kern else "deep" if "deep" in kern else "step%d" % per_launch)] = {
        "hbm_bytes_per_launch": ev["traffic_bytes"], "fetch_size_kib": fk, "write_size_kib": wk, "library_version": version,
        # the launches these bytes were counted on: bench.py prices `traffic_frac` on THEIR duration, not on the run's own
        "steps_per_launch": per_launch, "launch_us": round(stats["avg_ns"] / 1e3, 2),
        "source": "profiles/%s.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of the same bench.py command on %s, FETCH doubled per "
                  "MI355X_MICROARCH.md and checked on a 1 GiB copy)" % (tag, datetime.date.today()),
        "evidence": {k: v for k, v in ev.items() if k != "traffic_bytes"}}
    json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)

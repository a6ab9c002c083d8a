#!/usr/bin/env python3
"""tools/overlap_report.py <kernel_trace.csv> — how much of the halo exchange (RCCL kernels, or the halo_push /
halo_wait kernels of the peer transport; both reported under "xchg") and of the edge
launch runs concurrently with the interior launch of the multi-step kernels (d2q9_step2/3: edge = the small grid; d2q9_multi: edge = the launch with fewer tile rows)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = {"interior": [], "edge": [], "xchg": []}
for r in rows:
    name, st, en = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
    if "d2q9_step2" in name or "d2q9_step3" in name or "d2q9_step4" in name or "d2q9_multi" in name or "d2q9_deep" in name:
        iv["pending"] = iv.get("pending", []) + [(st, en, grid)]
    elif "nccl" in name.lower() or "rccl" in name.lower() or "halo_push" in name or "halo_wait" in name:
        iv["xchg"].append((st, en))
# the edge launch has 3*strips units (a few thousand threads), the interior launches many more; compact launch sets
# (peer transport) are ONE launch per set — its first workgroups are the edge units, which push the halo rows while the
# interior units of the same launch run: all launches have the same grid then and count as "interior"
p = iv.pop("pending")
gmin, gmax = min(g for _, _, g in p), max(g for _, _, g in p)
for st, en, g in p:
    (iv["edge"] if (g <= 2 * gmin and gmax > 2 * gmin) else iv["interior"]).append((st, en))
if not iv["edge"]:
    print("compact launch sets: one launch per set, the edge units and their pushes are inside it (no separate edge / push kernels)")
def overlap(a, others):
    tot = 0
    for (s, e) in others:
        tot += max(0, min(e, a[1]) - max(s, a[0]))
    return tot
n = len(iv["interior"])
span = max(e for _, e in iv["interior"]) - min(s for s, _ in iv["interior"])
print("launch sets: %d   wall span %.1f us   per set %.2f us" % (n, span / 1e3, span / 1e3 / n))
for k in ("interior", "edge", "xchg"):
    d = [e - s for s, e in iv[k]]
    print("%-9s kernels %5d   mean duration %8.2f us   total %9.1f us" % (k, len(d), sum(d) / max(1, len(d)) / 1e3, sum(d) / 1e3))
for k in ("edge", "xchg"):
    tot = sum(e - s for s, e in iv[k])
    ov = sum(overlap(a, iv["interior"]) for a in iv[k])
    print("%-5s time overlapped by an interior kernel: %.1f %%" % (k, 100.0 * ov / max(1, tot)))
busy = sum(e - s for s, e in iv["interior"])
print("interior kernels cover %.1f %% of the wall span" % (100.0 * busy / span))

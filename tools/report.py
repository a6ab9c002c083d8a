#!/usr/bin/env python3
"""tools/report.py — one table: library defaults on the reference's four inputs and on larger cavities."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lbm_amd
from conftest import input_files

K20M = {"128x128": 0.684, "128x256": 1.203, "256x256": 4.012, "1024x1024": 11.694}  # report.odt, seconds for the full run
rows = []
cases = [(s, None) for s in ("128x128", "128x256", "256x256", "1024x1024")] + [("cavity", n) for n in (2048, 4096, 8192, 16384)]
for name, n in cases:
    if n is None:
        p, ob = lbm_amd.read_inputs(*input_files(name))
        full = p.max_iters
        label = "input_%s" % name
    else:
        ob = np.zeros((n, n), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
        p = lbm_amd.make_params(n, n, 1, obstacles=ob)
        full = None
        label = "cavity %dx%d" % (n, n)
    steps = max(64, min(40000, int(2e10 / (p.nx * p.ny)) // 8 * 8))
    p.max_iters = steps * 3 + 64
    with lbm_amd.LBM(p, ob) as sim:
        sim.upload(None); sim.run(64)
        best = min(sim.run_timed(steps) for _ in range(3))
        kind = "d2q9_multi x%d" % sim.get_option("multistep") if sim.get_option("multistep") else ({0: "d2q9_step", 1: "d2q9_step2", 3: "d2q9_step3", 4: "d2q9_step4"}[sim.get_option("fuse")])
    us = best / steps * 1e3
    mlups = p.nx * p.ny * steps / best / 1e3
    ref = ""
    if full:
        ref = "%.3f s vs K20m %.3f s (%.0fx)" % (full * us * 1e-6, K20M[name], K20M[name] / (full * us * 1e-6))
    rows.append("| %s | %s | %.2f | %.0f | %.2f | %s |" % (label, kind, us, mlups, 72e-3 * mlups / 8000.0, ref))
    print(rows[-1], flush=True)
print()
print("| workload | kernel (auto) | us / step | MLUPS | 72 B x LUPS / 8 TB/s | full run, step loop |")
print("|---|---|---|---|---|---|")
print("\n".join(rows))

#!/usr/bin/env python3
"""tools/soak_deep.py [seconds] — repeats runs of d2q9_deep / d2q9_deep_twin (all depths, with and without the mailbox, slabs
with every transport a single process has) on random data and compares every result bit for bit with the single-step
kernel's: a hand-over or barrier race between the twins, or a stale halo row, would show as a sporadic mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(7)
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
cases = []
for (nx, ny, nsteps) in ((1024, 1024, 23), (2048, 512, 19), (512, 300, 37), (4096, 1024, 16), (8192, 1408, 23), (2048, 300, 24)):  # (8192x1408: slabs of 704 rows run chunk pairs)
    ob = (rng.random((ny, nx)) < 0.04).astype(np.int32)
    if (nx, ny) in ((2048, 300), (8192, 1408)):   # blocked cells in a band of rows and a band of columns only: free sweeps next to looking
        ob[40:, 256:] = 0                          # ones, and the strips of the column band (blocked in every row) balanced in one-round launches
    if (nx, ny) == (4096, 1024):                  # a cavity's side walls + a sprinkle: the two wall strips balanced
        ob[:] = (rng.random((ny, nx)) < 0.00002)
        ob[:, 0] = ob[:, -1] = 1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        sim.set_option("fuse", 0)
        sim.upload(cells0)
        sim.run(nsteps)
        ref, av_ref = sim.download()
    cases.append((nx, ny, nsteps, ob, cells0, p, ref, av_ref))
variants = [dict(fuse=8, pair=0), dict(fuse=8, pair=1, twin_steps=5), dict(fuse=8, pair=1, twin_steps=8), dict(fuse=7, pair=1, twin_steps=7),
            dict(fuse=6, pair=1, twin_steps=6), dict(fuse=8, pair=1, twin_steps=3), dict(fuse=8, pair=0, slabs=2), dict(fuse=8, pair=0, slabs=4)]
# round 3: with non-temporal stores (what the big grids run) launches of exactly 5 / 6 / 7 / 8 timesteps take the kernels
# instantiated per depth (steady form of the row loop); every variant once more in that configuration
variants += [dict(v, nt_stores=1, free_sweeps=1) for v in variants]   # (+ the map of blocked rows per strip consulted by every wave)
t0, runs, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    for (nx, ny, nsteps, ob, cells0, p, ref, av_ref) in cases:
        for v in variants:
            v = dict(v)
            slabs = v.pop("slabs", 1)
            if slabs > 1:
                lbm_amd.set_default("halo_depth", 8)
            try:
                with lbm_amd.LBM(p, ob, **(dict(devices=[0] * slabs) if slabs > 1 else {})) as sim:
                    sim.set_option("multistep", 0)
                    for k, val in v.items():
                        sim.set_option(k, val)
                    sim.upload(cells0)
                    sim.run(nsteps)
                    got, av = sim.download()
            finally:
                lbm_amd.set_default("halo_depth", 0)
            runs += 1
            if not np.array_equal(got, ref) or np.max(np.abs(av - av_ref) / np.abs(av_ref)) > 2e-6:
                bad += 1
                rows = np.argwhere(np.any(got != ref, axis=(0, 2))).ravel()
                print("MISMATCH %dx%d %s slabs %d: %d rows differ, first %s" % (nx, ny, v, slabs, rows.size, rows[:8]), flush=True)
    print("%.0f s: %d runs, %d mismatches" % (time.time() - t0, runs, bad), flush=True)
print("soak %s: %d runs, %d mismatches" % ("ok" if bad == 0 else "FAILED", runs, bad))
sys.exit(1 if bad else 0)

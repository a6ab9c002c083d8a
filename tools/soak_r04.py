#!/usr/bin/env python3
"""tools/soak_r04.py [seconds] — repeats runs of the three paths round 4 added and compares every result bit for bit with the
single-step kernel's: d2q9_resident (a late or torn exchange row, a wave that ran ahead of its neighbours' step words would
show as a sporadic mismatch), the five-step chunk pairs of row slabs with their edge pairs (peer stores between slabs of one
process), and the staged launch sets under RCCL (ring of one: staging blocks, stream wait-value, exchange beside the interior)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(11)
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1


def make(nx, ny, nsteps, blocked=0.04, walls=False):
    ob = (rng.random((ny, nx)) < blocked).astype(np.int32)
    ob[0, :] = ob[-1, :] = 0
    ob[ny // 2: ny // 2 + ny // 6, :] = 0      # a band of rows without blocked cells: both collision paths side by side
    if walls:
        ob[:, 0] = ob[:, -1] = 1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        sim.set_option("fuse", 0)
        sim.set_option("resident", 0)
        sim.upload(cells0)
        sim.run(nsteps)
        ref, av_ref = sim.download()
    return dict(nx=nx, ny=ny, nsteps=nsteps, ob=ob, cells0=cells0, p=p, ref=ref, av_ref=av_ref)


# (kind, case, creation keywords, creation defaults, options)
jobs = []
for (nx, ny, nsteps) in ((1024, 1024, 300), (512, 512, 97), (768, 768, 64), (1024, 512, 257), (128, 2048, 40), (896, 1024, 33),
                         (1000, 1000, 120), (900, 600, 77), (1020, 1536, 45), (260, 2048, 30)):   # (partly filled last waves; bands of six rows)
    c = make(nx, ny, nsteps)
    jobs.append(("resident", c, {}, {}, {"resident": 1}))
for (nx, ny, nsteps, slabs) in ((1024, 1024, 23, 2), (2048, 1024, 36, 4), (1024, 2000, 19, 3)):
    c = make(nx, ny, nsteps)
    jobs.append(("slab pairs x5, peer", c, dict(devices=[0] * slabs), {}, {}))
    jobs.append(("slab pairs x5, peer, in-kernel wait", c, dict(devices=[0] * slabs), {}, {"halo_sync": 2}))
for (nx, ny, nsteps, walls) in ((8192, 416, 23, True), (2048, 700, 23, False), (4096, 1024, 16, True)):
    c = make(nx, ny, nsteps, blocked=0.0003, walls=walls)
    jobs.append(("staged sets, rccl ring of one", c, "rccl", dict(force_halo=1, transport="rccl"), {}))

t0, runs, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    for (kind, c, kw, defaults, opts) in jobs:
        for k, v in defaults.items():
            lbm_amd.set_default(k, v)
        try:
            if kw == "rccl":
                kw_now = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id())
            else:
                kw_now = kw
            with lbm_amd.LBM(c["p"], c["ob"], **kw_now) as sim:
                for k, v in opts.items():
                    sim.set_option(k, v)
                if kind == "resident":
                    assert sim.get_option("resident") > 0
                elif kind.startswith("slab"):
                    assert sim.get_option("fuse") == 5 and sim.get_option("compact") == 1
                else:
                    assert sim.get_option("compact") == 1 and sim.get_option("transport") == 1
                sim.upload(c["cells0"])
                sim.run(c["nsteps"])
                got, av = sim.download()
        finally:
            for k in defaults:
                lbm_amd.set_default(k, 0)
        runs += 1
        if not np.array_equal(got, c["ref"]) or np.max(np.abs(av - c["av_ref"]) / np.abs(c["av_ref"])) > 2e-6:
            bad += 1
            rows = np.argwhere(np.any(got != c["ref"], axis=(0, 2))).ravel()
            print("MISMATCH %s %dx%d: %d rows differ, first %s" % (kind, c["nx"], c["ny"], rows.size, rows[:8]), flush=True)
    print("%.0f s: %d runs, %d mismatches" % (time.time() - t0, runs, bad), flush=True)
print("soak %s: %d runs, %d mismatches" % ("ok" if bad == 0 else "FAILED", runs, bad))
sys.exit(1 if bad else 0)

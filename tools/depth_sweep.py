#!/usr/bin/env python3
"""tools/depth_sweep.py [nx ny] — time of ONE launch of d2q9_deep by the number of timesteps it advances (2..6), next to
the four-step kernel: which part of a launch is the pass over the grid, which the arithmetic per level."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 8192)
ob = np.zeros((ny, nx), np.int32)
ob[0, :] = ob[-1, :] = 1
ob[:, 0] = ob[:, -1] = 1
p = lbm_amd.make_params(nx, ny, 4000, obstacles=ob)
DEEP = int(os.environ.get("DEEP", "6"))
for fuse, depths in ((DEEP, tuple(range(2, DEEP + 1))), (4, (4,)), (3, (3,))):
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        sim.set_option("fuse", fuse)
        sim.upload(None)
        sim.run(48)
        for L in depths:
            best = min(sim.run_timed(L) for _ in range(12))
            print("%dx%d fuse %d: launch of %d steps %8.1f us  %8.1f us/step  %7.1f GLUPS" % (
                nx, ny, fuse, L, best * 1e3, best * 1e3 / L, nx * ny * L / best / 1e6), flush=True)

#!/usr/bin/env python3
"""tools/run_case.py nx ny steps [key=value ...] — runs `steps` timesteps of a cavity grid with the given library options
(fuse=, chunk_rows=, variant=, nt_stores=, grid_blocks= ...; lanes_out= / halo_depth= / transport= / force_halo= are
creation defaults); profiling target for rocprofv3."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd
nx, ny, steps = (int(v) for v in sys.argv[1:4])
opts = dict(kv.split("=") for kv in sys.argv[4:])
for k in ("force_halo", "halo_depth", "transport", "lanes_out"):
    if k in opts:
        lbm_amd.set_default(k, int(opts.pop(k)))
ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
p = lbm_amd.make_params(nx, ny, steps + 8, obstacles=ob)
with lbm_amd.LBM(p, ob) as sim:
    for k, v in opts.items():
        sim.set_option(k, int(v))
    sim.upload(None)
    sim.run(8)
    ms = sim.run_timed(steps)
    print("%dx%d %s: %.5f ms/step, %.0f MLUPS" % (nx, ny, " ".join(sys.argv[4:]), ms / steps, nx * ny * steps / ms / 1e3))

#!/usr/bin/env python3
"""tools/run_ring.py nx ny steps [transport] — one slab that exchanges halos with itself (default force_halo; transport
rccl | peer | copy, default peer): profiling target that shows the edge launch / exchange / interior launch structure
of a rank of a multi-GPU run on a single GPU."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd
nx, ny, steps = (int(v) for v in sys.argv[1:4])
transport = sys.argv[4] if len(sys.argv) > 4 else "peer"
lbm_amd.set_default("force_halo", 1)
lbm_amd.set_default("transport", transport)
ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
p = lbm_amd.make_params(nx, ny, steps + 16, obstacles=ob)
kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) if transport == "rccl" else dict(devices=[0])
with lbm_amd.LBM(p, ob, **kw) as sim:
    sim.upload(None)
    sim.run(16)
    ms = sim.run_timed(steps)
    print("%dx%d ring of one (%s): %.2f us/step, %.0f MLUPS" % (nx, ny, transport, ms / steps * 1e3, nx * ny * steps / ms / 1e3))

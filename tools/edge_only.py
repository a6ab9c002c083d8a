"""ad-hoc: duration of the edge launch alone — a ring-of-one slab of 8 rows (4 + 4 edge rows, no interior) with halo depth 4"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ["LBM_FORCE_HALO"] = "1"; os.environ["LBM_HALO_DEPTH"] = "4"; os.environ["LBM_TRANSPORT"] = sys.argv[1] if len(sys.argv) > 1 else "copy"
import lbm_amd
nx, ny = 8192, 8
ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) if os.environ["LBM_TRANSPORT"] == "rccl" else dict(devices=[0])
with lbm_amd.LBM(p, ob, **kw) as sim:
    for fuse, per in ((3, 3), (4, 4), (1, 2)):
        sim.set_option("multistep", 0); sim.set_option("fuse", fuse)
        sim.upload(None); sim.run(48)
        steps = 960
        best = min(sim.run_timed(steps) for _ in range(3))
        print("8192x8 edge-only slab, transport %s, fuse=%d (reads back %d): %.2f us/step = %.1f us per launch set" % (os.environ["LBM_TRANSPORT"], fuse, sim.get_option("fuse"), best / steps * 1e3, best / steps * 1e3 * per), flush=True)

"""ad-hoc: slab mode (ring of one over RCCL): chunk pairs on/off for the interior launch"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ["LBM_FORCE_HALO"] = "1"; os.environ["LBM_TRANSPORT"] = "rccl"
import lbm_amd
for (nx, ny, steps) in [(8192, 1024, 480), (8192, 2048, 480), (8192, 4096, 240), (4096, 1024, 960), (2048, 1024, 960)]:
    ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob, rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) as sim:
        row = []
        for pair in (-1, 0, 1):
            sim.set_option("pair", pair)
            sim.upload(None); sim.run(48)
            best = min(sim.run_timed(steps) for _ in range(3))
            row.append("pair%2d(->%d) f%d units %-5d %6.2f us %6.0f" % (pair, sim.get_option("pair"), sim.get_option("fuse"), sim.get_option("fuse_units"), best / steps * 1e3, nx * ny * steps / best / 1e3))
        print("%5dx%-5d ring | " % (nx, ny) + " | ".join(row), flush=True)

"""ad-hoc: output lanes per strip of the two-step kernel (store alignment vs lane efficiency)"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
nx = ny = 8192
ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
for rnd in range(2):
    for lanes in (61, 60, 56, 52, 48, 32):
        os.environ["LBM_LANES_OUT"] = str(lanes)
        with lbm_amd.LBM(p, ob) as sim:
            sim.upload(None); sim.run(20)
            best = min(sim.run_timed(100) for _ in range(2))
            print("lanes_out=%d ms/step %.5f MLUPS %8.0f" % (lanes, best / 100, nx * ny * 100 / best / 1e3), flush=True)

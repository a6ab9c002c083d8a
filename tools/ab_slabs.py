"""ad-hoc: cost of the slab machinery on ONE GPU (several slabs on device 0, halos by D2D copies)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
def cavity(nx, ny):
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    return ob
for (nx, ny, steps) in [(8192, 8192, 100), (1024, 1024, 2000)]:
    ob = cavity(nx, ny)
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    for nslabs in (1, 2, 4, 8):
        for fuse in (1, 0):
            with lbm_amd.LBM(p, ob, devices=None if nslabs == 1 else [0] * nslabs) as sim:
                sim.set_option("fuse", fuse)
                sim.upload(None); sim.run(20)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d slabs=%d fuse=%d ms/step %.5f MLUPS %8.0f" % (nx, ny, nslabs, fuse, best / steps, nx * ny * steps / best / 1e3), flush=True)

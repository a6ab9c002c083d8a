import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(8192, 8192, 100), (4096, 4096, 400)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("fuse", 0)
        for variant in (3, 4, 1):
            sim.set_option("variant", variant)
            sim.upload(None); sim.run(20)
            best = min(sim.run_timed(steps) for _ in range(2))
            print("%dx%d single-step variant=%d us/step %.2f MLUPS %.0f GB/s %.0f" % (nx, ny, variant, best / steps * 1e3, nx * ny * steps / best / 1e3, 72e-6 * nx * ny * steps / best), flush=True)

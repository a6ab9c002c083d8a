import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(8192, 8192, 100), (4096, 4096, 400)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        for rnd in range(2):
            for ntl in (0, 1):
                sim.set_option("nt_loads", ntl)
                sim.upload(None); sim.run(20)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%dx%d nt_loads=%d ms/step %.5f MLUPS %8.0f" % (nx, ny, ntl, best / steps, nx * ny * steps / best / 1e3), flush=True)

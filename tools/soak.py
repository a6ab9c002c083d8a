"""ad-hoc: long runs for stability (finite values, mass drift, monotone spin-up of the cavity)"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(8192, 8192, 20000), (1024, 1024, 200000), (128, 128, 1000000)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, steps, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.upload(None)
        t0 = time.time(); sim.run(steps); sim.sync(); dt = time.time() - t0
        _, av = sim.download(cells=False)
        ux, uy, u, pr = sim.final_state()
        mass = float(pr.astype(np.float64).sum() * 3.0)
        print("%dx%d %d steps in %.2f s (%.0f MLUPS): finite=%s av[0]=%.3e av[-1]=%.3e mass/mass0=%.8f max|u|=%.3e" % (
            nx, ny, steps, dt, nx * ny * steps / dt / 1e6, bool(np.isfinite(av).all() and np.isfinite(pr).all()), av[0], av[-1],
            mass / (0.1 * nx * ny), float(u.max())), flush=True)

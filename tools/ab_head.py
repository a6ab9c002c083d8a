"""ad-hoc: state of HEAD on one box — two-step kernel with/without non-temporal loads, by grid size"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(8192, 8192, 100), (4096, 4096, 400), (2048, 2048, 1000), (1024, 1024, 2000)]:
  for chunk in (0,):
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("chunk_rows", chunk)
        for rnd in range(2):
            for (fuse, ntl, nts) in [(1, 2, -1)]:
                sim.set_option("fuse", fuse); sim.set_option("nt_loads", ntl); sim.set_option("nt_stores", nts)
                sim.upload(None); sim.run(20)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d chunk=%-2d fuse=%d nt_loads=%d nt_stores=%2d us/step %9.3f MLUPS %8.0f" % (nx, ny, chunk, fuse, ntl, nts, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

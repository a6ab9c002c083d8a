"""ad-hoc: chunk schedule of d2q9_step4 on big grids"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(8192, 8192, 240), (16384, 16384, 96), (8192, 4096, 480), (6144, 6144, 240), (4096, 8192, 480)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        del ob
        sim.set_option("multistep", 0)
        for rnd in range(2):
            for (fuse, chunk, cmin) in [(3, 0, 0), (4, 32, 8), (4, 32, 16), (4, 48, 8), (4, 48, 16), (4, 64, 8), (4, 64, 16), (4, 96, 16), (4, 128, 32)]:
                sim.set_option("fuse", fuse); sim.set_option("chunk_min", cmin); sim.set_option("chunk_rows", chunk)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d fuse=%d chunk=%-3d min=%-2d us/step %9.3f MLUPS %8.0f" % (nx, ny, fuse, chunk, cmin, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

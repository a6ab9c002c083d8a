"""ad-hoc: chunk schedule of d2q9_step3 (LDS windows) against d2q9_step2 over grid sizes"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
sizes = [(768, 768, 3840), (1024, 768, 3840), (1024, 1024, 1920), (1536, 1024, 1920), (2048, 1024, 1920), (2048, 2048, 960), (4096, 4096, 480), (8192, 1024, 480), (8192, 8192, 240)]
for (nx, ny, steps) in sizes:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        for rnd in range(2):
            for (fuse, chunk, cmin) in [(1, 0, 0), (3, 32, 6), (3, 16, 6), (3, 12, 4), (3, 10, 4), (3, 8, 4), (3, 8, 2), (3, 6, 2), (3, 4, 2)]:
                sim.set_option("fuse", fuse); sim.set_option("chunk_min", cmin); sim.set_option("chunk_rows", chunk)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d fuse=%d chunk=%-2d min=%-2d us/step %9.3f MLUPS %8.0f" % (nx, ny, fuse, chunk, cmin, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

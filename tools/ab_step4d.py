"""ad-hoc: where d2q9_step4 overtakes d2q9_step3 (default schedule), and its chunk schedule"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(1024, 768, 3840), (1024, 1024, 3840), (1536, 1024, 1920), (2048, 1024, 1920), (2048, 2048, 960), (3072, 2048, 960), (4096, 2048, 960), (4096, 4096, 480), (8192, 1024, 960), (8192, 8192, 240)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        for rnd in range(2):
            for (fuse, chunk, cmin) in [(3, 0, 0), (4, 128, 32), (4, 32, 8), (4, 16, 6), (4, 64, 16)]:
                sim.set_option("fuse", fuse); sim.set_option("chunk_min", cmin); sim.set_option("chunk_rows", chunk)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d fuse=%d chunk=%-3d min=%-2d units=%-6d us/step %9.3f MLUPS %8.0f" % (nx, ny, fuse, chunk, cmin, sim.get_option("fuse_units"), best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

"""ad-hoc: d2q9_step3 one-round schedule planned for 2 or 1 waves per SIMD on mid-size grids"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(768, 768, 3840), (1024, 768, 3840), (1024, 1024, 3840), (1536, 1024, 1920), (2048, 1024, 1920), (2048, 1536, 1920), (2048, 2048, 960), (3072, 2048, 960), (4096, 4096, 480)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("fuse", 3)
        for rnd in range(2):
            for (sw, chunk) in [(2, 0), (1, 0), (1, 32), (2, 32)]:
                sim.set_option("sched_waves", sw); sim.set_option("chunk_rows", chunk)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d sched_waves=%d chunk=%-2d units=%d us/step %9.3f MLUPS %8.0f" % (nx, ny, sw, chunk, sim.get_option("fuse_units"), best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

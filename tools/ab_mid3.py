"""ad-hoc: mid-size grids, d2q9_step3 variants planned for one wave per SIMD with two row-sets of loads in flight"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(768, 768, 3840), (1024, 768, 3840), (1024, 1024, 3840), (1536, 1024, 1920), (2048, 1024, 1920), (2048, 2048, 960)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("fuse", 3)
        for rnd in range(2):
            for (win, bufs, sw) in [(1, 1, 2), (1, 1, 1), (1, 2, 1), (0, 2, 1), (0, 1, 1)]:
                sim.set_option("windows", win); sim.set_option("load_bufs", bufs); sim.set_option("sched_waves", sw)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d windows=%d bufs=%d sched_waves=%d units=%-5d us/step %9.3f MLUPS %8.0f" % (nx, ny, win, bufs, sw, sim.get_option("fuse_units"), best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

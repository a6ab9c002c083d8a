"""ad-hoc: mid-size grids, d2q9_step3 with LDS windows: one row-set of loads in flight (180 VGPRs) against two (253 VGPRs
with plain loads: still two waves per SIMD)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(1024, 768, 3840), (1024, 1024, 3840), (1536, 1024, 1920), (2048, 1024, 1920), (2048, 2048, 960), (4096, 4096, 480), (8192, 8192, 240)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("fuse", 3); sim.set_option("windows", 1)
        for rnd in range(2):
            for (bufs, ntl) in [(1, 2), (1, 0), (2, 0), (1, 1)]:
                sim.set_option("load_bufs", bufs); sim.set_option("nt_loads", ntl)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d bufs=%d nt_loads=%d us/step %9.3f MLUPS %8.0f" % (nx, ny, bufs, ntl, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

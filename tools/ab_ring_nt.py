"""ad-hoc: source-load mode of d2q9_step3 in slab mode (ring of one over RCCL)"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ["LBM_FORCE_HALO"] = "1"; os.environ["LBM_TRANSPORT"] = "rccl"
import lbm_amd
for (nx, ny, steps) in [(8192, 1024, 480), (8192, 2048, 480), (4096, 1024, 960)]:
    ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob, rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) as sim:
        for rnd in range(2):
            for ntl in (0, 2, 1):
                sim.set_option("nt_loads", ntl)
                sim.upload(None); sim.run(48)
                best = min(sim.run_timed(steps) for _ in range(3))
                print("%5dx%-5d ring of one nt_loads=%d us/step %8.2f MLUPS %8.0f" % (nx, ny, ntl, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

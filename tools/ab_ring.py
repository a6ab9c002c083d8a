"""ad-hoc: cost of a launch set with RCCL halo exchange, measured on one GPU with a ring of one
(LBM_FORCE_HALO=1): the slab sizes an 8-GPU run gives each rank"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
os.environ["LBM_FORCE_HALO"] = "1"
for (nx, ny, steps) in [(1024, 128, 4000), (1024, 256, 4000), (1024, 512, 4000), (8192, 1024, 400)]:
    ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    for transport in ("rccl", "copy"):
        os.environ["LBM_TRANSPORT"] = transport
        kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) if transport == "rccl" else dict(devices=[0])
        with lbm_amd.LBM(p, ob, **kw) as sim:
            for (fuse, ms) in ((-1, -1), (1, 0), (0, 0)):
                sim.set_option("fuse", fuse); sim.set_option("multistep", ms)
                sim.upload(None); sim.run(40)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d ring-of-1 %s fuse=%2d multistep=%2d (effective %d/%d) us/step %.2f MLUPS %8.0f" % (nx, ny, transport, fuse, ms, sim.get_option("fuse"), sim.get_option("multistep"), best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

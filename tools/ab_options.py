"""ad-hoc A/B of library options on the GPU box (not a test)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd

def cavity(nx, ny):
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    return ob

for (nx, ny, steps) in [(8192, 8192, 100), (4096, 4096, 400), (2048, 2048, 1000), (1024, 1024, 2000)]:
    ob = cavity(nx, ny)
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        configs = [("fuse", 0, 0, 0)] + [("fuse", 1, cr, cm) for cr in (6, 8, 12, 16, 32) for cm in (2, 4) if cr <= ny and cm <= cr]
        for rnd in range(2):
            for (_, fuse, cr, cm) in configs:
                sim.set_option("fuse", fuse)
                if fuse:
                    sim.set_option("chunk_min", cm)
                    sim.set_option("chunk_rows", cr)
                sim.upload(None)
                sim.run(20)
                ms = sim.run_timed(steps)
                mlups = nx * ny * steps / (ms * 1e-3) / 1e6
                print("%5dx%-5d fuse=%d chunk=%-4d min=%d units=%d ms/step %.5f  MLUPS %8.0f  GB/s(72B) %6.0f" % (nx, ny, fuse, cr, cm, sim.get_option("fuse_units"), ms / steps, mlups, mlups * 72e-3), flush=True)

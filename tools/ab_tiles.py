"""ad-hoc: d2q9_multi tile shape by grid size"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny) in [(128, 128), (128, 256), (256, 256), (384, 384), (512, 512), (1024, 512)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 400000, obstacles=ob)
    steps = 8000
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 8)
        row = []
        for shape in (0, 1, 2, -1):
            sim.set_option("tile_shape", shape)
            sim.upload(None); sim.run(96)
            best = min(sim.run_timed(steps) for _ in range(3))
            row.append("shape%2d %6.3f us" % (shape, best / steps * 1e3))
        print("%5dx%-5d | " % (nx, ny) + " | ".join(row), flush=True)

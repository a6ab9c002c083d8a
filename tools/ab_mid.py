"""ad-hoc: kernel choice vs grid size (where the auto thresholds sit)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny) in [(384, 384), (512, 512), (768, 512), (1024, 512), (768, 768), (1024, 768), (1024, 1024), (1536, 1024), (2048, 2048), (4096, 4096)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 400000, obstacles=ob)
    steps = max(240, int(4e8 / (nx * ny)) // 24 * 24)
    with lbm_amd.LBM(p, ob) as sim:
        row = []
        for (ms, fuse) in [(0, 0), (0, 1), (0, 3), (0, 4), (8, 0)]:
            sim.set_option("fuse", fuse); sim.set_option("multistep", ms)
            sim.upload(None); sim.run(96)
            best = min(sim.run_timed(steps) for _ in range(3))
            row.append("ms%d/f%d %7.2f us %7.0f" % (ms, fuse, best / steps * 1e3, nx * ny * steps / best / 1e3))
        print("%5dx%-5d | " % (nx, ny) + " | ".join(row), flush=True)

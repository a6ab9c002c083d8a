"""ad-hoc: 300K..1M cells — which kernel"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(512, 512, 3840), (768, 512, 3840), (1024, 512, 3840), (768, 768, 3840), (1280, 512, 3840), (1024, 768, 3840), (1024, 1024, 3840)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        row = []
        for (ms, fuse, sw) in [(8, 0, 0), (0, 1, 0), (0, 3, 2), (0, 3, 1), (0, 4, 0)]:
            sim.set_option("multistep", ms); sim.set_option("fuse", fuse); sim.set_option("sched_waves", sw)
            sim.upload(None); sim.run(48)
            best = min(sim.run_timed(steps) for _ in range(3))
            row.append("ms%d/f%d/sw%d %6.2f us %6.0f" % (ms, fuse, sw, best / steps * 1e3, nx * ny * steps / best / 1e3))
        print("%5dx%-5d | " % (nx, ny) + " | ".join(row), flush=True)

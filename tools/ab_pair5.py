"""ad-hoc: auto pairing policy vs forced on/off"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(1024, 512, 3840), (1024, 1024, 3840), (2048, 1024, 1920), (2048, 2048, 960), (4096, 2048, 960), (4096, 4096, 480), (8192, 1024, 960), (8192, 2048, 480), (8192, 4096, 240), (8192, 8192, 240)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        row = []
        for pair in (-1, 0, 1):
            sim.set_option("pair", pair)
            sim.upload(None); sim.run(48)
            best = min(sim.run_timed(steps) for _ in range(3))
            row.append("pair%2d(->%d) f%d units %-5d %6.2f us %6.0f" % (pair, sim.get_option("pair"), sim.get_option("fuse"), sim.get_option("fuse_units"), best / steps * 1e3, nx * ny * steps / best / 1e3))
        print("%5dx%-5d | " % (nx, ny) + " | ".join(row), flush=True)

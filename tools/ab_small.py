"""ad-hoc: small-grid step time, vec 4 vs 1"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lbm_amd
from conftest import input_files
for size, steps in [("128x128", 8000), ("128x256", 8000), ("256x256", 8000), ("1024x1024", 2000)]:
    p, ob = lbm_amd.read_inputs(*input_files(size))
    p.max_iters = 200000
    with lbm_amd.LBM(p, ob) as sim:
        for vec in (4, 1):
            for fuse in (0, 1):
                if fuse and not (p.nx >= 256): continue
                sim.set_option("fuse", fuse); sim.set_option("vec", vec)
                sim.upload(None); sim.run(100)
                best = min(sim.run_timed(steps) for _ in range(3))
                print("%-10s vec=%d fuse=%d us/step %.3f MLUPS %8.0f" % (size, vec, fuse, best / steps * 1e3, p.nx * p.ny * steps / best / 1e3), flush=True)

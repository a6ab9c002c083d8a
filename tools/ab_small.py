"""ad-hoc: small-grid step time by kernel choice (shipped inputs)"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lbm_amd
from conftest import input_files
for size, steps in [("128x128", 8000), ("128x256", 8000), ("256x256", 8000), ("1024x1024", 2000)]:
    p, ob = lbm_amd.read_inputs(*input_files(size))
    p.max_iters = 200000
    with lbm_amd.LBM(p, ob) as sim:
        for (ms, fuse) in [(0, 0), (0, 1), (2, 0), (4, 0), (8, 0)]:
            if fuse and p.nx < 256: continue
            sim.set_option("fuse", fuse); sim.set_option("multistep", ms)
            sim.upload(None); sim.run(96)
            best = min(sim.run_timed(steps) for _ in range(3))
            print("%-10s multistep=%d fuse=%d us/step %.3f MLUPS %8.0f" % (size, ms, fuse, best / steps * 1e3, p.nx * p.ny * steps / best / 1e3), flush=True)

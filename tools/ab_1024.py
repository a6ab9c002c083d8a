"""ad-hoc: 1024x1024 (shipped input) chunk schedule of the two-step kernel + alternatives"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lbm_amd
from conftest import input_files
p, ob = lbm_amd.read_inputs(*input_files("1024x1024"))
p.max_iters = 400000
steps = 4000
with lbm_amd.LBM(p, ob) as sim:
    sim.set_option("multistep", 0)
    for rnd in range(2):
        for (cr, cm) in [(0, 0), (8, 4), (3, 3)]:
            sim.set_option("fuse", 1); sim.set_option("chunk_min", cm); sim.set_option("chunk_rows", cr)
            sim.upload(None); sim.run(40)
            best = min(sim.run_timed(steps) for _ in range(2))
            print("1024x1024 chunk=%d min=%d units=%d us/step %.3f MLUPS %8.0f" % (cr, cm, sim.get_option("fuse_units"), best / steps * 1e3, 1024 * 1024 * steps / best / 1e3), flush=True)
    sim.set_option("fuse", 0); sim.upload(None); sim.run(40)
    best = min(sim.run_timed(steps) for _ in range(2)); print("single-step us/step %.3f" % (best / steps * 1e3))
    sim.set_option("multistep", 8); sim.upload(None); sim.run(40)
    best = min(sim.run_timed(steps) for _ in range(2)); print("multistep8 us/step %.3f" % (best / steps * 1e3))

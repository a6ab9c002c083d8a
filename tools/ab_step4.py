"""ad-hoc: four timesteps per launch (d2q9_step4) — bit-identity against single steps and speed against d2q9_step3"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
rng = np.random.default_rng(5)
for (nx, ny, nsteps, chunk) in [(256, 37, 9, 5), (512, 64, 8, 0), (1024, 50, 13, 7), (2048, 16, 4, 16), (260, 33, 11, 4), (1024, 1024, 10, 0)]:
    ob = (rng.random((ny, nx)) < 0.08).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    res = {}
    for fuse in (0, 4):
        with lbm_amd.LBM(p, ob) as sim:
            sim.set_option("multistep", 0); sim.set_option("fuse", fuse); sim.set_option("chunk_rows", chunk)
            sim.upload(cells0); sim.run(nsteps); res[fuse] = sim.download() + (sim.get_option("fuse"),)
    same = np.array_equal(res[0][0], res[4][0])
    dav = float(np.max(np.abs(res[0][1] - res[4][1]) / np.abs(res[0][1])))
    print("%5dx%-5d %2d steps chunk %2d: fuse option reads back %d, grids identical %s, av_vels max rel diff %.1e" % (nx, ny, nsteps, chunk, res[4][2], same, dav), flush=True)
for (nx, ny, steps) in [(8192, 8192, 240), (4096, 4096, 480), (2048, 2048, 960), (1024, 1024, 3840)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        for rnd in range(2):
            for (fuse, chunk, cmin) in [(3, 0, 0), (4, 0, 0), (4, 24, 8), (4, 32, 8), (4, 12, 6)]:
                sim.set_option("fuse", fuse); sim.set_option("chunk_min", cmin); sim.set_option("chunk_rows", chunk)
                sim.upload(None); sim.run(24)
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d fuse=%d chunk=%-2d min=%-2d us/step %9.3f MLUPS %8.0f" % (nx, ny, fuse, chunk, cmin, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

"""ad-hoc: d2q9_step3p (chunk pairs share their start-up rows) against d2q9_step3 — identity and speed"""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
rng = np.random.default_rng(5)
for (nx, ny, nsteps, chunk) in [(256, 37, 9, 5), (512, 64, 6, 0), (1024, 50, 13, 7), (2048, 16, 4, 16), (260, 33, 11, 4), (1024, 1024, 10, 0), (256, 8, 3, 0)]:
    ob = (rng.random((ny, nx)) < 0.08).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    res = {}
    for (fuse, pair) in ((0, 0), (3, 1)):
        with lbm_amd.LBM(p, ob) as sim:
            sim.set_option("multistep", 0); sim.set_option("fuse", fuse); sim.set_option("pair", pair); sim.set_option("chunk_rows", chunk)
            sim.upload(cells0); sim.run(nsteps); res[fuse] = sim.download()
    same = np.array_equal(res[0][0], res[3][0])
    dav = float(np.max(np.abs(res[0][1] - res[3][1]) / np.abs(res[0][1])))
    print("%5dx%-5d %2d steps chunk %2d: grids identical %s, av_vels max rel diff %.1e" % (nx, ny, nsteps, chunk, same, dav), flush=True)
for (nx, ny, steps) in [(1024, 512, 3840), (768, 768, 3840), (1024, 768, 3840), (1024, 1024, 3840), (1536, 1024, 1920), (2048, 1024, 1920), (2048, 2048, 960), (8192, 8192, 240)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("fuse", 3)
        row = []
        for (pair, sw) in [(0, 0), (1, 0), (1, 2), (1, 1)]:
            sim.set_option("pair", pair); sim.set_option("sched_waves", sw)
            sim.upload(None); sim.run(48)
            best = min(sim.run_timed(steps) for _ in range(3))
            row.append("pair%d/sw%d %6.2f us %6.0f" % (pair, sw, best / steps * 1e3, nx * ny * steps / best / 1e3))
        print("%5dx%-5d | " % (nx, ny) + " | ".join(row), flush=True)

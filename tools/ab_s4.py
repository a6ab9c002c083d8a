import sys, zlib
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
rng = np.random.default_rng(5)
for (nx, ny, nsteps, chunk) in [(256, 37, 9, 5), (1024, 50, 13, 7), (260, 33, 11, 4), (2048, 300, 8, 128)]:
    ob = (rng.random((ny, nx)) < 0.08).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    res = {}
    for fuse in (0, 4):
        with lbm_amd.LBM(p, ob) as sim:
            sim.set_option("multistep", 0); sim.set_option("fuse", fuse); sim.set_option("chunk_rows", chunk)
            sim.upload(cells0); sim.run(nsteps); res[fuse] = sim.download()
    print("%5dx%-5d identical to single steps: %s" % (nx, ny, np.array_equal(res[0][0], res[4][0])), flush=True)
for (nx, ny, steps) in [(8192, 8192, 240), (4096, 4096, 480), (16384, 16384, 96), (8192, 1024, 960)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        del ob
        sim.set_option("fuse", 4)
        sim.upload(None); sim.run(24)
        best = min(sim.run_timed(steps) for _ in range(3))
        print("%5dx%-5d fuse=%d us/step %9.3f MLUPS %8.0f" % (nx, ny, sim.get_option("fuse"), best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

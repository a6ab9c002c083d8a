import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, chunk, nsteps) in [(256, 37, 5, 2), (256, 40, 5, 2), (256, 16, 8, 2), (256, 37, 37, 2), (512, 64, 32, 2)]:
    rng = np.random.default_rng(1)
    ob = (rng.random((ny, nx)) < 0.08).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    res = []
    for fuse in (0, 1):
        with lbm_amd.LBM(p, ob) as sim:
            sim.set_option("fuse", fuse)
            if fuse:
                sim.set_option("chunk_rows", chunk)
            sim.upload(cells0)
            sim.run(nsteps)
            res.append(sim.download())
    d = res[0][0] != res[1][0]
    print(nx, ny, chunk, "mismatch cells:", int(d.sum()), "av", res[0][1], res[1][1])
    if d.any():
        k, y, x = np.nonzero(d)
        print("  planes", sorted(set(k.tolist())), "rows", sorted(set(y.tolist()))[:40], "cols min/max", x.min(), x.max(), "n distinct cols", len(set(x.tolist())))
        i = 0
        print("  e.g.", k[i], y[i], x[i], res[0][0][k[i], y[i], x[i]], res[1][0][k[i], y[i], x[i]])

import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lbm_amd
rng = np.random.default_rng(21)
nx, ny = 512, 96
ob = (rng.random((ny, nx)) < 0.05).astype(np.int32)
ob[0, :] = ob[-1, :] = 0
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
def run(nsteps, opts, ring, devices=(0,)):
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    lbm_amd.set_default("force_halo", 1 if ring else 0)
    lbm_amd.set_default("transport", "peer" if ring else "auto")
    kw = dict(devices=list(devices)) if ring else {}
    with lbm_amd.LBM(p, ob, **kw) as sim:
        for k, v in opts.items():
            sim.set_option(k, v)
        info = {k: sim.get_option(k) for k in ("fuse", "multistep", "pair")}
        if ring:
            info["compact"] = sim.get_option("compact"); info["halo_depth"] = sim.get_option("halo_depth")
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()
    return got, info
for nsteps in (4, 8, 12):
    ref, _ = run(nsteps, {"fuse": 0, "multistep": 0}, False)
    for label, opts, devs in (("compact f4", {"fuse": 4, "multistep": 0}, (0,)), ("compact f4 chunk16", {"fuse": 4, "multistep": 0, "chunk_rows": 16, "chunk_min": 16}, (0,)),
                              ("two-stream f4", {"fuse": 4, "multistep": 0, "compact": 0}, (0,)), ("compact f3", {"fuse": 3, "multistep": 0}, (0,)),
                              ("compact f4 2 slabs", {"fuse": 4, "multistep": 0}, (0, 0)), ("compact f4 pair0->two-stream", {"fuse": 4, "multistep": 0, "pair": 0}, (0,))):
        got, info = run(nsteps, opts, True, devs)
        bad = np.argwhere(np.any(got != ref, axis=(0, 2))).ravel()
        print("%2d steps %-28s %s -> %s" % (nsteps, label, info, "identical" if bad.size == 0 else "DIFFERS in %d rows: %s ... %s" % (bad.size, bad[:8], bad[-8:])), flush=True)

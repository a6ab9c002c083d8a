"""Helper of test_gpu_parity.py::test_rccl_transport_self_ring — run as a subprocess (so that a hang in
the communication layer is a test failure, not a stuck test session).  One rank, LBM_FORCE_HALO=1: the
slab is its own north and south neighbour and every halo exchange is an RCCL send/recv to self."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd

transport = sys.argv[1]  # "rccl" or "copy"
rng = np.random.default_rng(21)
nx, ny, nsteps = 512, 96, 23
ob = (rng.random((ny, nx)) < 0.05).astype(np.int32)
ob[0, :] = ob[-1, :] = 0
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)

with lbm_amd.LBM(p, ob) as sim:
    sim.set_option("fuse", 0)
    sim.set_option("multistep", 0)
    sim.upload(cells0)
    sim.run(nsteps)
    ref, av_ref = sim.download()
    re_ref = sim.reynolds()

os.environ["LBM_FORCE_HALO"] = "1"
os.environ["LBM_TRANSPORT"] = transport
for (fuse, ms) in ((0, 0), (1, 0), (3, 0), (4, 0), (0, 8), (0, 5)):  # 1 / 2 / 3 / 4 / 8 / 5 timesteps per launch set (halo depth 8)
    kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) if transport == "rccl" else dict(devices=[0])
    with lbm_amd.LBM(p, ob, **kw) as sim:
        assert sim.get_option("transport") == (1 if transport == "rccl" else 2)
        sim.set_option("fuse", fuse)
        sim.set_option("multistep", ms)
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()   # rank mode: av_vels go through ncclAllReduce
        re = sim.reynolds()
    assert np.array_equal(got, ref), "state differs (transport %s, fuse %d, multistep %d)" % (transport, fuse, ms)
    assert np.max(np.abs(av - av_ref) / av_ref) < 2e-6
    assert abs(re / re_ref - 1) < 1e-5
print("self-ring ok:", transport)

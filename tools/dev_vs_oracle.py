#!/usr/bin/env python3
"""tools/dev_vs_oracle.py [steps ...] — how far the HIP timestep sits from the fp32 oracle (max relative deviation of every
distribution and of av_vels) on the shipped inputs and on a random case: the margins behind the parity tests' tolerances
(2e-5 on cells, 1e-4 on av_vels)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

steps = [int(v) for v in sys.argv[1:]] or [11, 1000, 4000]
orc = Oracle("f32", omp=True)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def case(name, p, ob, cells0, opts):
    for n in steps:
        p.max_iters = n
        po = orc.make_params(p.nx, p.ny, n, p.reynolds_dim, p.density, p.accel, p.omega)
        orc.set_obstacles(po, ob)
        ref = cells0.copy() if cells0 is not None else orc.init_cells(po)
        av_ref = orc.run(po, ref, ob, n)
        with lbm_amd.LBM(p, ob) as sim:
            for k, v in opts.items():
                sim.set_option(k, v)
            sim.upload(cells0)
            sim.run(n)
            got, av = sim.download()
        print("%-34s %5d steps: cells %.2e  av_vels %.2e  mass/oracle-1 %+.2e" % (
            name, n, rel(got, ref), rel(av, av_ref), got.astype(np.float64).sum() / ref.astype(np.float64).sum() - 1), flush=True)


for size in ("128x128", "128x256", "256x256", "1024x1024"):
    p, ob = lbm_amd.read_inputs(os.path.join(ROOT, "inputs", "input_%s.params" % size), os.path.join(ROOT, "inputs", "obstacles_%s.dat" % size))
    case(size + " (default kernel)", p, ob, None, {})
rng = np.random.default_rng(1)
nx, ny = 512, 256
ob = (rng.random((ny, nx)) < 0.08).astype(np.int32)
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
c0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
case("random 512x256", lbm_amd.make_params(nx, ny, 1, obstacles=ob), ob, c0, {})
case("random 512x256 single step", lbm_amd.make_params(nx, ny, 1, obstacles=ob), ob, c0, {"fuse": 0, "multistep": 0})

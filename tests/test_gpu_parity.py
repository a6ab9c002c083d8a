"""GPU parity: the HIP timestep (through the C ABI) against the fp32 CPU oracle on the same
inputs.  Tolerances: the GPU contracts a*b+c into FMAs and uses v_rcp_f32 /
v_sqrt_f32 (1 ulp), the oracle is built with -ffp-contract=off, so after n steps the states differ
by accumulated fp32 rounding only: <= 2e-5 relative on every distribution for n <= 1000 and
<= 1e-4 relative on av_vels.  The acceptance gate of the reference itself is 1 % (check.py)."""
import numpy as np
import pytest

from conftest import SIZES, input_files

pytestmark = pytest.mark.gpu

RTOL_CELLS = 2e-5
RTOL_AV = 1e-4


def max_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def run_both(lbm, orc, p_gpu, obstacles, cells0, nsteps, **kw):
    po = orc.make_params(p_gpu.nx, p_gpu.ny, p_gpu.max_iters, p_gpu.reynolds_dim, p_gpu.density, p_gpu.accel,
                         p_gpu.omega)
    orc.set_obstacles(po, obstacles)
    ref = cells0.copy()
    av_ref = orc.run(po, ref, obstacles, nsteps)
    with lbm.LBM(p_gpu, obstacles, **kw) as sim:
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()
    return got, av, ref, av_ref


@pytest.mark.parametrize("size", SIZES[:3])
@pytest.mark.parametrize("nsteps", [1, 2, 10, 1001])
def test_shipped_inputs_vs_oracle(lbm, oracle_f32_omp, size, nsteps):
    p, obst = lbm.read_inputs(*input_files(size))
    p.max_iters = nsteps
    cells0 = oracle_f32_omp.init_cells(oracle_f32_omp.make_params(p.nx, p.ny, nsteps, 10, p.density, p.accel, p.omega))
    got, av, ref, av_ref = run_both(lbm, oracle_f32_omp, p, obst, cells0, nsteps)
    assert max_rel(got, ref) < RTOL_CELLS
    assert max_rel(av, av_ref) < RTOL_AV

"""GPU parity: the HIP timestep (through the C ABI) against the fp32 CPU oracle on the same
inputs.  Tolerances: the GPU contracts a*b+c into FMAs, regroups the equilibrium
terms and uses v_rcp_f32 / v_sqrt_f32 (1 ulp); the oracle is built with -ffp-contract=off.  After n
steps the states differ by accumulated fp32 rounding only: <= 2e-5 relative on every distribution
for n <= ~1000 and <= 1e-4 relative on av_vels.  The acceptance gate of the reference itself is
1 % on av_vels and pressure (check/check.py) and is applied to the full-length runs below."""
import io
import os
import subprocess

import numpy as np
import pytest

from conftest import (ROOT, SIZES, generated_final_state, golden_cols, golden_path, input_files, write_av_vels,
                      write_final_state)

pytestmark = pytest.mark.gpu

RTOL_CELLS = 2e-5
RTOL_AV = 1e-4
# explicit kernel choices (library defaults pick by grid size: LDS multi-step kernel up to ~512x512,
# two-steps-per-launch kernel from ~1024x768, one step per launch in between and as the odd last step)
SINGLE = {"fuse": 0, "multistep": 0}
FUSED2 = {"fuse": 1, "multistep": 0}
FUSED3 = {"fuse": 3, "multistep": 0}
FUSED4 = {"fuse": 4, "multistep": 0}


def max_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def oracle_params(orc, p, obstacles):
    po = orc.make_params(p.nx, p.ny, p.max_iters, p.reynolds_dim, p.density, p.accel, p.omega)
    orc.set_obstacles(po, obstacles)
    return po


def run_gpu(lbm, p, obstacles, cells0, nsteps, options=None, **kw):
    with lbm.LBM(p, obstacles, **kw) as sim:
        for k, v in (options or {}).items():
            sim.set_option(k, v)
        sim.upload(cells0)
        sim.run(nsteps)
        return sim.download()


def random_case(rng, nx, ny, blocked=0.08):
    """random obstacles (never on the accelerated row only by chance) + perturbed positive state"""
    ob = (rng.random((ny, nx)) < blocked).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    return ob, cells


# ---- shipped inputs, a few steps, exact-arithmetic comparison -------------------------------------

@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("nsteps", [1, 2, 11, 1000])
def test_shipped_inputs_vs_oracle(lbm, oracle_f32_omp, size, nsteps):
    if size == "1024x1024" and nsteps == 1000:
        nsteps = 200
    p, obst = lbm.read_inputs(*input_files(size))
    p.max_iters = nsteps
    po = oracle_params(oracle_f32_omp, p, obst)
    ref = oracle_f32_omp.init_cells(po)
    cells0 = ref.copy()
    av_ref = oracle_f32_omp.run(po, ref, obst, nsteps)
    got, av = run_gpu(lbm, p, obst, cells0, nsteps)
    assert max_rel(got, ref) < RTOL_CELLS
    assert max_rel(av, av_ref) < RTOL_AV


# ---- every kernel variant, edge cases of the geometry ------------------------------------------------

@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("nt", [0, 1])
def test_all_load_variants_agree_bitwise(lbm, oracle_f32, variant, nt):
    """the four neighbour-fetch strategies and both store flavours compute identical bits"""
    rng = np.random.default_rng(7)
    nx, ny, nsteps = 512, 24, 9
    ob, cells0 = random_case(rng, nx, ny)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    base, av_base = run_gpu(lbm, p, ob, cells0, nsteps, dict(SINGLE, variant=1, nt_stores=0))
    got, av = run_gpu(lbm, p, ob, cells0, nsteps, dict(SINGLE, variant=variant, nt_stores=nt))
    assert np.array_equal(got, base) and np.array_equal(av, av_base)
    po = oracle_params(oracle_f32, p, ob)
    ref = cells0.copy()
    av_ref = oracle_f32.run(po, ref, ob, nsteps)
    assert max_rel(got, ref) < RTOL_CELLS and max_rel(av, av_ref) < RTOL_AV


@pytest.mark.parametrize("nx,ny,chunk", [(256, 8, 0), (256, 37, 5), (512, 64, 32), (1024, 50, 7), (2048, 16, 16), (260, 33, 4),
                                         (8192, 24, 8)])
@pytest.mark.parametrize("nsteps", [2, 3, 9])
def test_two_steps_per_launch_equals_single_steps(lbm, oracle_f32_omp, nx, ny, chunk, nsteps):
    """d2q9_step2 (two timesteps per launch, intermediate state in registers) performs the same per-cell
    arithmetic as two launches of d2q9_step: the states must agree bit for bit, av_vels up to summation
    order; odd step counts end with one single-step launch; both agree with the oracle"""
    rng = np.random.default_rng(nx + 7 * ny + nsteps)
    ob, cells0 = random_case(rng, nx, ny)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        sim.set_option("fuse", 1)
        sim.set_option("chunk_rows", chunk)
        assert sim.get_option("fuse") == 1
        sim.upload(cells0)
        sim.run(nsteps)
        fused, av_fused = sim.download()
    assert np.array_equal(fused, single)
    assert max_rel(av_fused, av_single) < 2e-6
    po = oracle_params(oracle_f32_omp, p, ob)
    ref = cells0.copy()
    av_ref = oracle_f32_omp.run(po, ref, ob, nsteps)
    assert max_rel(fused, ref) < RTOL_CELLS and max_rel(av_fused, av_ref) < RTOL_AV


@pytest.mark.parametrize("nx,ny,bh", [(128, 6, 2), (128, 4, 2), (256, 64, 2), (512, 512, 2), (1024, 512, 2), (1024, 1024, 4), (512, 2048, 4), (128, 4100, 4),
                                      (384, 64, 2), (640, 512, 2), (768, 768, 4), (896, 256, 2), (1024, 1536, 6), (512, 3072, 6), (1024, 1200, 6),
                                      # widths that are no multiple of 128: the band's last wave is partly filled
                                      (1000, 1000, 4), (260, 512, 2), (132, 64, 2), (900, 600, 4), (1020, 1536, 6), (516, 96, 2)])
@pytest.mark.parametrize("nsteps,split", [(1, 0), (2, 0), (7, 3), (23, 0), (300, 0)])
def test_resident_kernel_equals_single_steps(lbm, nx, ny, bh, nsteps, split):
    """d2q9_resident: all steps of an lbm_run in ONE launch with the grid in registers — bands of 2 or 4 full-width rows, one to
    eight waves across, x neighbours through LDS, y neighbours through exchange rows in memory behind per-wave step words.
    Random obstacles (open top and bottom rows: the y wrap between the last and the first band carries flow), an obstacle-free
    band of rows (both collision paths), 300 steps = two launches (the ring of partial sums holds 256), split runs; the state
    must equal single steps bit for bit"""
    rng = np.random.default_rng(900 + nx + ny)
    ob, cells0 = random_case(rng, nx, ny, blocked=0.06)
    ob[0, :] = 0
    ob[-1, :] = 0
    ob[ny // 2: ny // 2 + max(1, ny // 8), :] = 0
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as sim:
        sim.set_option("resident", 1)
        assert sim.get_option("resident") == bh
        sim.upload(cells0)
        if split:
            sim.run(split)
            sim.sync()
        sim.run(nsteps - split)
        got, av = sim.download()
    assert np.array_equal(one, got)
    assert max_rel(av, av_one) < 2e-6


def test_two_resident_contexts_side_by_side(lbm):
    """a resident launch needs every band on the device at once; two contexts of one process that are run without a sync in between
    (each on its own stream) must not wait for each other's workgroups: the library orders their launches.  Both results equal
    single steps, and the pair takes about as long as the two runs one after the other (not the 30 s of a timed-out wait)"""
    import time
    rng = np.random.default_rng(77)
    nx, ny, nsteps = 1024, 1024, 600
    ob, cells0 = random_case(rng, nx, ny, blocked=0.02)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as a, lbm.LBM(p, ob) as b:
        assert a.get_option("resident") == 4 and b.get_option("resident") == 4
        a.upload(cells0)
        b.upload(cells0)
        a.sync()
        b.sync()
        t0 = time.perf_counter()
        a.run(nsteps)       # asynchronous: three launches of up to 256 steps each
        b.run(nsteps)
        ga, ava = a.download()
        gb, avb = b.download()
        dt = time.perf_counter() - t0
    assert np.array_equal(ga, one) and np.array_equal(gb, one)
    assert max_rel(ava, av_one) < 2e-6 and max_rel(avb, av_one) < 2e-6
    assert dt < 5.0


@pytest.mark.parametrize("nx,ny", [(128, 128), (128, 256), (256, 256), (3, 3), (5, 4), (33, 17), (100, 70), (130, 31), (512, 48)])
@pytest.mark.parametrize("T,nsteps", [(1, 3), (2, 7), (3, 8), (8, 8), (8, 21), (5, 16)])
def test_lds_multistep_equals_single_steps(lbm, oracle_f32_omp, nx, ny, T, nsteps):
    """d2q9_multi (T timesteps per launch on LDS tiles with 2T-cell redundant halos; tiles that hang over the grid
    edge, regions that wrap around tiny grids several times, step counts that are no multiple of T) performs the same
    per-cell arithmetic as T launches of d2q9_step: bit-identical states, av_vels equal up to summation order"""
    rng = np.random.default_rng(nx * 31 + ny + T)
    ob, cells0 = random_case(rng, nx, ny, blocked=0.08 if nx * ny > 30 else 0.0)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as sim:
        sim.set_option("multistep", T)
        assert sim.get_option("multistep") == T
        sim.upload(cells0)
        sim.run(nsteps)
        multi, av_multi = sim.download()
    assert np.array_equal(multi, single)
    assert np.max(np.abs(av_multi - av_single)) <= 2e-6 * np.max(np.abs(av_single)) + 1e-12
    po = oracle_params(oracle_f32_omp, p, ob)
    ref = cells0.copy()
    av_ref = oracle_f32_omp.run(po, ref, ob, nsteps)
    assert max_rel(multi, ref) < RTOL_CELLS
    assert np.max(np.abs(av_multi - av_ref)) <= RTOL_AV * np.max(np.abs(av_ref)) + 1e-12


def test_default_kernel_choice_by_grid_size(lbm):
    """auto policy: LDS multi-step kernel for launch-bound grids, two-step kernel in between, three- and four-step kernels
    for bandwidth-bound ones, the deep window kernel above 300K cells — as chunk pairs (d2q9_deep_twin), with up to five steps
    per launch below 3M cells and up to eight from there on; grids of fewer than 32 rows, which the deep kernel does not take, fall
    back to the two- / three-step kernels; and, round 4, the resident kernel (all steps of a launch with the grid in registers)
    from 200K to 1.5M cells where the grid is a multiple of 4 cells (128 to 1024) wide and its bands of 2, 4 or 6 rows all fit the chip at once"""
    expect = {(128, 128): (8, 0, 8), (256, 256): (8, 0, 8), (512, 512): "resident 2", (640, 512): "resident 2", (768, 512): "resident 2", (1024, 512): "resident 2", (1152, 512): (0, 8, 5),
              (16384, 24): (0, 1, 2), (32768, 24): (0, 3, 3), (512, 384): (8, 0, 8),
              (768, 768): "resident 4", (1024, 1024): "resident 4", (1024, 1536): "resident 6", (1024, 2048): (0, 8, 5), (1000, 1000): "resident 4", (1002, 1000): (0, 0, 1), (1024, 1028): (0, 8, 5), (1536, 1024): (0, 8, 5), (2048, 1024): (0, 8, 5), (2048, 2048): (0, 8, 8),
              (3072, 2048): (0, 8, 8), (4096, 2048): (0, 8, 8), (8192, 1024): (0, 8, 8), (128, 8192): "resident 4", (128, 8200): (0, 0, 1)}
    for (nx, ny), want in expect.items():
        ob = np.zeros((ny, nx), np.int32)
        with lbm.LBM(lbm.make_params(nx, ny, 4, obstacles=ob), ob) as sim:
            if isinstance(want, str):
                assert sim.get_option("resident") == int(want.split()[1]) and sim.get_option("multistep") == 0, (nx, ny)
                sim.run(4)
                continue
            ms, fuse, per_launch = want
            assert sim.get_option("resident") == 0, (nx, ny)
            assert (sim.get_option("multistep"), sim.get_option("fuse") if not ms else 0) == (ms, fuse), (nx, ny)
            assert sim.get_option("launch_steps") == per_launch, (nx, ny)
            if fuse == 8:   # one slab without halo rows: the deep window kernel always runs as chunk pairs (d2q9_deep_twin)
                assert sim.get_option("pair") == 1, (nx, ny)
            sim.run(4)  # and it runs


@pytest.mark.parametrize("nx,ny", [(3, 3), (5, 4), (30, 17), (132, 40), (256, 3), (260, 7), (1024, 5), (64, 300)])
def test_ragged_sizes(lbm, oracle_f32, nx, ny):
    """nx not a multiple of 4 (scalar kernel), of 256 (no wave-level modes), tiny and thin grids;
    the reference itself only accepts nx % 128 == 0 and power-of-two work-group counts"""
    rng = np.random.default_rng(nx * 1000 + ny)
    ob, cells0 = random_case(rng, nx, ny, blocked=0.1 if nx * ny > 20 else 0.0)
    nsteps = 7
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    po = oracle_params(oracle_f32, p, ob)
    ref = cells0.copy()
    av_ref = oracle_f32.run(po, ref, ob, nsteps)
    got, av = run_gpu(lbm, p, ob, cells0, nsteps)
    assert max_rel(got, ref) < RTOL_CELLS
    assert np.max(np.abs(av - av_ref)) <= RTOL_AV * np.max(np.abs(av_ref)) + 1e-12


def test_all_cells_blocked_and_none_blocked(lbm, oracle_f32):
    nx, ny, nsteps = 64, 16, 5
    rng = np.random.default_rng(3)
    _, cells0 = random_case(rng, nx, ny)
    for ob in (np.zeros((ny, nx), np.int32), np.ones((ny, nx), np.int32)):
        p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
        if ob.all():
            p.free_cells_inv = 1.0  # 1/0 in the reference; any finite value, av_vels must be exactly 0
        po = oracle_params(oracle_f32, p, ob)
        po.free_cells_inv = p.free_cells_inv
        ref = cells0.copy()
        av_ref = oracle_f32.run(po, ref, ob, nsteps)
        got, av = run_gpu(lbm, p, ob, cells0, nsteps)
        if ob.all():
            assert np.array_equal(got, ref) and np.all(av == 0.0)  # pure bounce-back permutation: bit-exact
        else:
            assert max_rel(got, ref) < RTOL_CELLS and max_rel(av, av_ref) < RTOL_AV


def test_step_counts_and_repeated_runs(lbm, oracle_f32):
    """odd step counts (the reference reads back a fixed buffer: only even counts are right there),
    a run split over several lbm_run calls, more steps than one reduction batch (256)"""
    rng = np.random.default_rng(11)
    nx, ny = 128, 32
    ob, cells0 = random_case(rng, nx, ny)
    total = 300
    p = lbm.make_params(nx, ny, total, obstacles=ob)
    whole, av_whole = run_gpu(lbm, p, ob, cells0, total)
    with lbm.LBM(p, ob) as sim:
        sim.upload(cells0)
        for n in (1, 2, 0, 254, 3, 40):
            sim.run(n)
        assert sim.steps_done == total
        parts, av_parts = sim.download()
        with pytest.raises(lbm.LBMError):
            sim.run(1)  # av_vels record is full
    # states are bit-identical however the run is cut; av_vels only up to summation order (the cut changes
    # which kernel / which sub-step of a multi-step launch computes a given step)
    assert np.array_equal(whole, parts) and max_rel(av_parts, av_whole) < 2e-6
    po = oracle_params(oracle_f32, p, ob)
    ref = cells0.copy()
    av_ref = oracle_f32.run(po, ref, ob, total)
    assert max_rel(whole, ref) < RTOL_CELLS and max_rel(av_whole, av_ref) < RTOL_AV
    for odd in (1, 3, 7):
        p.max_iters = odd
        got, _ = run_gpu(lbm, p, ob, cells0, odd)
        ref = cells0.copy()
        oracle_f32.run(oracle_params(oracle_f32, p, ob), ref, ob, odd)
        assert max_rel(got, ref) < RTOL_CELLS


def test_device_side_initial_state_equals_upload(lbm, oracle_f32):
    p, obst = lbm.read_inputs(*input_files("128x128"))
    p.max_iters = 5
    cells0 = oracle_f32.init_cells(oracle_params(oracle_f32, p, obst))
    a, ava = run_gpu(lbm, p, obst, cells0, 5)
    b, avb = run_gpu(lbm, p, obst, None, 5)
    assert np.array_equal(a, b) and np.array_equal(ava, avb)


def test_mass_conservation_and_rest_state(lbm):
    """total_density (d2q9-bgk.c:754-770) is constant; without forcing the rest state is a fixed point
    and av_vels is exactly zero (pairwise momentum differences)"""
    p, obst = lbm.read_inputs(*input_files("128x256"))
    p.max_iters = 400
    with lbm.LBM(p, obst) as sim:
        sim.upload(None)
        c0, _ = sim.download(av_vels=False)
        sim.run(400)
        c1, _ = sim.download(av_vels=False)
    # fp32 rounding of 9 stores per cell and step: <= half an ulp (6e-8) each, observed drift ~1e-8 per step
    assert abs(c1.astype(np.float64).sum() / c0.astype(np.float64).sum() - 1.0) < 400 * 6e-8
    p.accel = 0.0
    p.max_iters = 50
    got, av = run_gpu(lbm, p, obst, None, 50)
    assert np.all(av == 0.0)
    assert max_rel(got, c0) < 1e-6


# ---- output stage -------------------------------------------------------------------------------------

def test_final_state_and_reynolds(lbm, oracle_f32):
    p, obst = lbm.read_inputs(*input_files("128x128"))
    p.max_iters = 500
    with lbm.LBM(p, obst) as sim:
        sim.upload(None)
        sim.run(500)
        cells, _ = sim.download()
        ux, uy, u, pr = sim.final_state()
        re = sim.reynolds()
    po = oracle_params(oracle_f32, p, obst)
    rux, ruy, ru, rpr = oracle_f32.final_fields(po, cells, obst)  # same state -> only the output arithmetic differs
    assert max_rel(pr, rpr) < 1e-6
    assert np.max(np.abs(ux - rux)) < 1e-6 * np.max(np.abs(rux)) and np.max(np.abs(uy - ruy)) < 1e-6 * np.max(np.abs(ruy))
    assert np.max(np.abs(u - ru)) < 1e-6 * np.max(ru)
    assert np.all(pr[obst != 0] == np.float32(p.density) * np.float32(1.0 / 3.0)) and np.all(u[obst != 0] == 0)
    assert abs(re / oracle_f32.reynolds(po, cells, obst) - 1.0) < 1e-4  # the oracle sums 16k terms in fp32


def test_final_state_into_page_locked_memory(lbm):
    """lbm_host_alloc / lbm_host_free (the read-back targets of the C host and of bench.py's reference-rule leg): the same
    columns land in page-locked memory as in pageable memory, argument errors are refused, freeing twice is harmless"""
    import ctypes
    p, obst = lbm.read_inputs(*input_files("128x256"))
    p.max_iters = 60
    with lbm.LBM(p, obst) as sim:
        sim.upload(None)
        sim.run(60)
        plain = sim.final_state()
        buf = lbm.HostBuffer((4, p.ny, p.nx))
        buf.array[:] = -1.0
        pinned = sim.final_state(out=buf.array)
        assert all(np.array_equal(a, b) for a, b in zip(plain, pinned))
        assert np.array_equal(buf.array[3], plain[3]) and not np.any(buf.array == -1.0)
        buf.close()
        buf.close()
    lib = lbm.load_library()
    assert lib.lbm_host_alloc(None, 64) != 0 and b"bad argument" in lib.lbm_last_error()
    ptr = ctypes.c_void_p()
    assert lib.lbm_host_alloc(ctypes.byref(ptr), 0) != 0
    assert lib.lbm_host_free(None) == 0


# ---- row partition on one GPU (several slabs on device 0, halos by device-to-device copies) ----------

@pytest.mark.parametrize("nslabs,ny", [(2, 50), (3, 50), (8, 50), (2, 16), (4, 67), (5, 128), (2, 260)])
@pytest.mark.parametrize("mode", ["single", "fused2", "fused3", "multi8", "multi3", "auto"])
def test_row_slabs_equal_single_slab(lbm, nslabs, ny, mode):
    """several slabs on one GPU (halo rows exchanged by device-to-device copies) against one slab, with one,
    two and up to eight timesteps per launch set (halo depth 2 or 8); 37 steps = full launch sets + a remainder"""
    opts = {"single": SINGLE, "fused2": FUSED2, "fused3": FUSED3, "multi8": {"multistep": 8}, "multi3": {"multistep": 3},
            "auto": {}}[mode]
    rng = np.random.default_rng(5)
    nx, nsteps = 256, 37
    ob, cells0 = random_case(rng, nx, ny)
    ob[0, :] = 0
    ob[-1, :] = 0  # open top/bottom: the y wrap-around between the last and the first slab carries flow
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    many, av_many = run_gpu(lbm, p, ob, cells0, nsteps, opts, devices=[0] * nslabs)
    assert np.array_equal(one, many)                 # per-cell arithmetic is identical
    assert max_rel(av_many, av_one) < 2e-6           # only the summation order of av_vels differs


@pytest.mark.parametrize("transport", ["copy", "rccl", "peer", "rccl_deep", "rccl_deep twin5"])
def test_transport_self_ring(transport):
    """the halo-exchange machinery with ONE slab that is its own ring neighbour, once per transport: rccl = every
    exchange is ncclSend/ncclRecv (to self) on a communicator made by ncclCommInitRank and av_vels go through
    ncclAllReduce — the code path of the one-process-per-GPU launch; peer = halo_push kernel + flag words with both
    consumer-side waits, and a context that switches between RCCL and peer stores mid-run; rccl_deep = the staged launch sets of
    the deep window kernel under RCCL (one launch per set, pushes into local staging blocks, exchange behind a stream wait-value)"""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_self_ring.py")] + transport.split(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "self-ring ok" in r.stdout


@pytest.mark.parametrize("world,nx,ny,nsteps,fuse,multistep,sync", [(2, 512, 96, 23, 3, 0, 0), (2, 512, 96, 23, 8, 0, 0), (4, 2048, 256, 30, 8, 0, 0), (3, 256, 150, 29, 0, 8, 0),
                                                                   (4, 2048, 64, 14, 4, 0, 1), (2, 256, 24, 11, 0, 0, 1),
                                                                   (4, 1024, 256, 203, 0, 8, 2), (2, 300, 40, 37, 0, 5, 2),
                                                                   (4, 2048, 256, 30, 4, 0, 2), (3, 1024, 300, 25, 3, 0, 2),
                                                                   # slabs of 704 / 1400 rows: the interior runs as chunk pairs (d2q9_deep_twin<..., PUSH>),
                                                                   # one round with a late pair / the tapered multi-round schedule
                                                                   (2, 8192, 1408, 23, 8, 0, 0), (3, 8192, 4200, 16, 8, 0, 2),
                                                                   # ... with a cavity's side walls: balanced wall strips + free sweeps on every rank
                                                                   (2, 8192, 1408, 23, 8, 0, "walls"), (3, 8192, 2112, 16, 8, 0, "walls"),
                                                                   # slabs with five halo rows: d2q9_deep_twin<5, ..., PUSH>, edge chunk pairs (the shipped 1024x1024
                                                                   # input over 2 and 4 ranks has these shapes)
                                                                   (2, 1024, 1024, 23, 5, -1, 0), (4, 1024, 1024, 36, 5, -1, 0), (3, 2048, 1500, 16, 5, -1, 2)])
def test_peer_transport_between_processes(world, nx, ny, nsteps, fuse, multistep, sync):
    """the peer transport across PROCESS boundaries: `world` processes share the one GPU, each owns a row slab, maps
    its neighbours' grids and flag words through HIP IPC, pushes its edge rows into them and waits on its own flags
    (wait kernel, hipStreamWaitValue32, or the edge tiles of the consuming d2q9_multi launch themselves); descriptors travel over torch.distributed/gloo.  The assembled state must
    equal the single-slab run bit for bit.  (RCCL cannot do this on one GPU: it refuses two ranks per device.)"""
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_ipc_ring.py"),
                        str(nx), str(ny), str(nsteps), str(fuse), str(multistep)] + (["0", "walls"] if sync == "walls" else [str(sync)]),
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if r.returncode != 0 or "ipc-ring ok" not in r.stdout:
        print(r.stdout[-3000:])
        print("\n".join(ln for ln in r.stderr.splitlines() if "Gloo" not in ln)[-6000:])
    assert r.returncode == 0 and "ipc-ring ok" in r.stdout


@pytest.mark.parametrize("nslabs,ny", [(2, 50), (3, 50), (2, 16), (4, 67), (5, 128), (2, 260)])
def test_row_slabs_four_steps_per_launch(lbm, nslabs, ny, halo_defaults):
    """slabs with halo depth 4 (what slabs of 2M cells and more get) and d2q9_step4 on edge + interior launches:
    bit-identical to one slab; 37 steps = nine full launch sets + one leftover step"""
    halo_defaults(halo_depth=4)
    rng = np.random.default_rng(6)
    nx, nsteps = 256, 37
    ob, cells0 = random_case(rng, nx, ny)
    ob[0, :] = 0
    ob[-1, :] = 0
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0] * nslabs) as sim:
        sim.set_option("multistep", 0)
        sim.set_option("fuse", 4)
        # slabs thinner than 8 rows fall back to halo depth 2 and the two-step kernel
        assert sim.get_option("fuse") == (4 if ny // nslabs >= 8 else 1)
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


@pytest.mark.parametrize("nslabs,nx,ny,depth", [(2, 256, 64, 8), (3, 512, 130, 8), (2, 1024, 260, 6), (4, 260, 200, 7), (5, 256, 160, 8)])
def test_row_slabs_deep_kernel(lbm, nslabs, nx, ny, depth, halo_defaults):
    """slabs with halo depth 8 (what slabs of 3M cells and more get) and d2q9_deep on edge + interior launches: bit-identical
    to one slab; 37 steps = launch sets of 8, 8, 7, 7, 7 (depth 8); a halo depth below the kernel's limit caps the sets"""
    halo_defaults(halo_depth=8 if depth != 7 else 7)
    rng = np.random.default_rng(60 + nslabs)
    nsteps = 37
    ob, cells0 = random_case(rng, nx, ny)
    ob[0, :] = 0
    ob[-1, :] = 0
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0] * nslabs) as sim:
        sim.set_option("multistep", 0)
        sim.set_option("fuse", 8 if depth == 7 else depth)
        assert sim.get_option("fuse") == depth and sim.get_option("halo_depth") == (7 if depth == 7 else 8)
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


@pytest.mark.parametrize("pair,halo_sync,nsteps,transport", [(-1, 0, 23, "peer"), (-1, 2, 16, "peer"), (0, 0, 23, "peer"), (-1, 0, 23, "copy")])
def test_row_slabs_deep_kernel_chunk_pairs(lbm, pair, halo_sync, nsteps, transport, halo_defaults):
    """row slabs big enough for the interior's chunk PAIRS (d2q9_deep_twin<..., PUSH>: one edge workgroup per strip — bottom
    edge rows on wave 0, top edge rows on wave 1 — then the interior's pairs, a late pair last): 8192x1408 over two slabs of 704
    rows (12 pairs + 1 late pair per strip), launch sets of 8 + 8 + 7 steps (per-depth kernels) or two of 8, peer stores with
    the wait kernel or with the edge waves polling the flags themselves; with device-to-device copies (two-stream launch sets: an
    edge launch beside the interior launch) the interior stays on the lone kernel — pairs there were measured at 200 against 292
    GLUPS, their 40-KB workgroups crowd out the edge launch and the exchange; bit-identical to single steps on one slab"""
    halo_defaults(transport=transport)
    rng = np.random.default_rng(77)
    nx, ny = 8192, 1408
    ob = (rng.random((ny, nx)) < 0.01).astype(np.int32)
    ob[300:500, :] = 0           # a band without blocked cells: both collision paths
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0, 0]) as sim:
        sim.set_option("pair", pair)
        sim.set_option("halo_sync", halo_sync)
        assert sim.get_option("fuse") == 8 and sim.get_option("halo_depth") == 8
        assert sim.get_option("transport") == {"peer": 3, "copy": 2}[transport]
        assert sim.get_option("pair") == (1 if (pair != 0 and transport == "peer") else 0)
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


def test_row_slabs_large_fused(lbm):
    """2048x512 over 4 slabs with the two-step kernel's edge/interior split and tapered schedule"""
    rng = np.random.default_rng(9)
    nx, ny, nsteps = 2048, 512, 11
    ob, cells0 = random_case(rng, nx, ny, blocked=0.02)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    for opts in (FUSED2, FUSED3, {}):  # {} = auto: three steps per launch on 2048x128 slabs? no - 8 steps on LDS tiles
        many, av_many = run_gpu(lbm, p, ob, cells0, nsteps, opts, devices=[0, 0, 0, 0])
        assert np.array_equal(one, many) and max_rel(av_many, av_one) < 2e-6


@pytest.mark.parametrize("transport,fuse,launch_steps", [("peer", 5, 5), ("copy", 4, 4)])
def test_row_slabs_of_2m_cells_carry_five_halo_rows(lbm, transport, fuse, launch_steps, halo_defaults):
    """8192x512 over 2 slabs: each slab holds 2M cells, so the library picks halo depth 5 by itself — the five-step chunk pairs
    in compact launch sets (peer stores), d2q9_step4 on two streams (copies)"""
    halo_defaults(transport=transport)
    rng = np.random.default_rng(10)
    nx, ny, nsteps = 8192, 512, 14
    ob, cells0 = random_case(rng, nx, ny, blocked=0.02)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0, 0]) as sim:
        assert sim.get_option("halo_depth") == 5 and sim.get_option("fuse") == fuse and sim.get_option("multistep") == 0
        assert sim.get_option("launch_steps") == launch_steps
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many) and max_rel(av_many, av_one) < 2e-6


@pytest.mark.parametrize("nslabs,nx,ny", [(1, 1024, 512), (2, 1024, 640), (4, 2048, 1024), (3, 1024, 2000), (2, 256, 2600), (8, 1024, 2560)])
@pytest.mark.parametrize("nsteps,halo_sync,nt,paths", [(23, 0, -1, -1), (16, 2, -1, -1), (11, 0, 1, -1), (7, 0, -1, 0), (1, 0, -1, -1)])
def test_row_slabs_five_step_chunk_pairs(lbm, nslabs, nx, ny, nsteps, halo_sync, nt, paths, halo_defaults):
    """row slabs of 300K to 3M cells (what the shipped 1024x1024 input gives 2 GPUs): five halo rows exchanged (of five stored above
    540K cells per slab, of the eight the LDS tiles' range stores below), compact launch sets
    of d2q9_deep_twin<5, ..., PUSH> — the five rows at either end of a slab are a chunk pair of their own (2 + 3 rows), every edge
    wave pushes its rows into the ring neighbour, the interior is one round of chunk pairs; launch sets of 5 + 5 + 5 + 4 + 4
    (per-depth kernel and the any-depth one), 4 x 4, 4 + 4 + 3, 4 + 3 and a single step; one slab as its own neighbour, uneven
    splits (667 / 666 rows), a narrow grid (3 strips), 8 slabs; wait kernel and in-kernel wait; bit-identical to single steps"""
    if nslabs == 1:
        halo_defaults(force_halo=1)
    rng = np.random.default_rng(500 + nslabs)
    ob, cells0 = random_case(rng, nx, ny, blocked=0.03)
    ob[0, :] = 0
    ob[-1, :] = 0  # open top/bottom: the y wrap-around between the last and the first slab carries flow
    ob[ny // 3: ny // 3 + 40, :] = 0  # a band without blocked cells: both collision paths
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0] * nslabs) as sim:
        sim.set_option("halo_sync", halo_sync)
        sim.set_option("nt_stores", nt)
        sim.set_option("obst_paths", paths)
        assert sim.get_option("halo_depth") == (5 if nx * (ny // nslabs) > 540 * 1024 else 8)
        assert sim.get_option("fuse") == 5 and sim.get_option("multistep") == 0
        assert sim.get_option("compact") == 1 and sim.get_option("pair") == 1 and sim.get_option("launch_steps") == 5
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


def test_row_slabs_split_runs_and_shipped_geometry(lbm, oracle_f32_omp):
    p, obst = lbm.read_inputs(*input_files("128x256"))  # periodic in y: rows 0 and 255 are open
    p.max_iters = 120
    with lbm.LBM(p, obst, devices=[0, 0, 0, 0]) as sim:
        assert sim.get_option("nslabs") == 4
        sim.upload(None)
        sim.run(1)
        sim.run(119)
        got, av = sim.download()
        ux, uy, u, pr = sim.final_state()
    po = oracle_params(oracle_f32_omp, p, obst)
    ref = oracle_f32_omp.init_cells(po)
    av_ref = oracle_f32_omp.run(po, ref, obst, 120)
    assert max_rel(got, ref) < RTOL_CELLS and max_rel(av, av_ref) < RTOL_AV
    assert max_rel(pr, oracle_f32_omp.final_fields(po, ref, obst)[3]) < RTOL_CELLS


# ---- acceptance: full-length runs of the four shipped inputs through the reference's checker --------

def check_outputs(tmp_path, size, av, fields, obst):
    from check.check import run_check
    fs, avf = str(tmp_path / "final_state.dat"), str(tmp_path / "av_vels.dat")
    write_final_state(fs, obst, *fields)
    write_av_vels(avf, av)
    ref_av = golden_path("%s.av_vels.dat" % size, tmp_path)
    if size in ("128x128", "128x256"):
        ref_fs = golden_path("%s.final_state.dat" % size, tmp_path)
    else:  # the reference repository lacks these two files; regenerated with the pinned fp64 oracle
        ref_fs = generated_final_state(size, str(tmp_path / ("ref_%s.final_state.dat" % size)))
    out = io.StringIO()
    code, avd, fsd = run_check(ref_av, ref_fs, avf, fs, 1.0, out)
    assert code == 0, out.getvalue()
    return avd, fsd


@pytest.mark.parametrize("size", SIZES)
def test_full_run_passes_reference_checker(lbm, tmp_path, size):
    p, obst = lbm.read_inputs(*input_files(size))
    with lbm.LBM(p, obst) as sim:
        sim.upload(None)
        sim.run(p.max_iters)
        _, av = sim.download(cells=False)
        fields = sim.final_state()
        re = sim.reynolds()
    avd, fsd = check_outputs(tmp_path, size, av, fields, obst)
    # fp32 drift against the fp64 golden files stays well inside the 1 % gate (<= 0.24 %)
    assert abs(avd["max_diff_pcnt"]) < 0.5 and abs(fsd["max_diff_pcnt"]) < 0.5
    ref_re = {"128x128": 9.763598020526, "128x256": 37.18483826704, "256x256": 10.07703420252,
              "1024x1024": 3.377417654904}[size]
    assert abs(re / ref_re - 1.0) < 5e-3


@pytest.mark.parametrize("mode", ["slabs8", "slabs8_copy", "ring_rccl", "ring_peer"])
def test_partitioned_full_run_passes_checker(tmp_path, mode):
    """BASELINE config 4 as far as one GPU goes: the shipped 1024x1024 input, all 20000 steps, row-partitioned — over
    8 slabs (128 rows each, halo depth 8, d2q9_multi; peer stores or device-to-device copies) and as ONE rank that is
    its own ring neighbour over RCCL send/recv and over peer stores — through the reference's checker against the
    golden av_vels and the fp64-oracle final state, and bit-identical to the undivided run"""
    import sys
    from check.check import run_check
    outs = {}
    for m in ("single", mode):
        d = tmp_path / m
        d.mkdir()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_partitioned_run.py"), m, "1024x1024", str(d)],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "partitioned run ok" in r.stdout, r.stdout + r.stderr
        outs[m] = np.load(d / "state.npz")
    ref_av = golden_path("1024x1024.av_vels.dat", tmp_path)
    ref_fs = generated_final_state("1024x1024", str(tmp_path / "ref_final_state.dat"))
    out = io.StringIO()
    code, avd, fsd = run_check(ref_av, ref_fs, str(tmp_path / mode / "av_vels.dat"), str(tmp_path / mode / "final_state.dat"), 1.0, out)
    assert code == 0, out.getvalue()
    assert abs(avd["max_diff_pcnt"]) < 0.5 and abs(fsd["max_diff_pcnt"]) < 0.5
    one, many = outs["single"], outs[mode]
    for k in ("ux", "uy", "u", "pr"):
        assert np.array_equal(one[k], many[k]), k           # same per-cell arithmetic: identical final state
    assert max_rel(many["av"], one["av"]) < 2e-6            # velocity sums: summation order only
    assert abs(float(many["re"]) / float(one["re"]) - 1.0) < 1e-5


def test_8192x8192_cavity_over_8_slabs_equals_one_slab(lbm):
    """BASELINE config 5's grid, row-partitioned the way an 8-GPU run partitions it (8 slabs of 8192x1024 = 8M cells:
    halo depth 8, d2q9_deep on edge and interior launches, peer stores), all on the one GPU: 19 timesteps (launch sets
    of 7, 6 and 6 steps) bit-identical to the undivided grid"""
    nx = ny = 8192
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    nsteps = 19
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    with lbm.LBM(p, ob) as sim:
        sim.upload(None)
        sim.run(nsteps)
        one, av_one = sim.download()
    with lbm.LBM(p, ob, devices=[0] * 8) as sim:
        assert sim.get_option("nslabs") == 8 and sim.get_option("fuse") == 8 and sim.get_option("halo_depth") == 8
        assert sim.get_option("transport") == 3
        sim.upload(None)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


def run_check_cli(tmp_path, size):
    """check/check.py as a program (the reference's `make check`, Makefile:26-27) on the files in tmp_path"""
    import sys
    ref_av = golden_path("%s.av_vels.dat" % size, tmp_path)
    if size in ("128x128", "128x256"):
        ref_fs = golden_path("%s.final_state.dat" % size, tmp_path)
    else:  # the reference repository lacks these two files; regenerated with the pinned fp64 oracle
        ref_fs = generated_final_state(size, str(tmp_path / ("ref_%s.final_state.dat" % size)))
    return subprocess.run([sys.executable, os.path.join(ROOT, "check", "check.py"),
                           "--ref-av-vels-file=" + ref_av, "--ref-final-state-file=" + ref_fs,
                           "--av-vels-file=" + str(tmp_path / "av_vels.dat"),
                           "--final-state-file=" + str(tmp_path / "final_state.dat")], capture_output=True, text=True)


@pytest.mark.parametrize("size", SIZES)
def test_c_host_acceptance_all_shipped_inputs(tmp_path, size):
    """north_star's acceptance gate through the DROP-IN itself: ./d2q9-bgk.exe <params> <obstacles> (full length) ->
    final_state.dat + av_vels.dat -> check/check.py as a program, for each of the four shipped inputs; 1024x1024 a second
    time row-partitioned over four slabs (LBM_DEVICES=0,0,0,0: the multi-GPU data path on one GPU), byte-identical
    final state"""
    exe = os.path.join(ROOT, "d2q9-bgk.exe")
    r = subprocess.run([exe, *input_files(size)], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for label in ("==done==", "Reynolds number:\t\t", "Elapsed time:\t\t\t", "Elapsed user CPU time:\t\t",
                  "Elapsed system CPU time:\t"):
        assert label in r.stdout
    c = run_check_cli(tmp_path, size)
    assert c.returncode == 0 and "Both tests passed!" in c.stdout, c.stdout + c.stderr
    nx, ny = (int(v) for v in size.split("x"))
    first = open(tmp_path / "final_state.dat", "rb").read()
    assert first.count(b"\n") == nx * ny
    ref_re = {"128x128": 9.763598020526, "128x256": 37.18483826704, "256x256": 10.07703420252,
              "1024x1024": 3.377417654904}[size]
    got_re = float([ln for ln in r.stdout.splitlines() if ln.startswith("Reynolds number:")][0].split()[2])
    assert abs(got_re / ref_re - 1.0) < 5e-3
    if size == "1024x1024":
        env = dict(os.environ, LBM_DEVICES="0,0,0,0")
        r = subprocess.run([exe, *input_files(size)], cwd=tmp_path, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert open(tmp_path / "final_state.dat", "rb").read() == first     # same per-cell arithmetic on every slab
        c = run_check_cli(tmp_path, size)
        assert c.returncode == 0 and "Both tests passed!" in c.stdout, c.stdout + c.stderr


def test_c_host_writer_and_slab_environment(tmp_path):
    """the host's output writer (line format, thread count) and its LBM_DEVICES / LBM_NGPUS environment"""
    exe = os.path.join(ROOT, "d2q9-bgk.exe")
    r = subprocess.run([exe, *input_files("128x128")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the multi-threaded writer produces the same bytes as a single thread, in the reference's line format
    first = open(tmp_path / "final_state.dat", "rb").read()
    import re
    assert re.fullmatch(rb"\d+ \d+ -?\d\.\d{12}E[+-]\d\d -?\d\.\d{12}E[+-]\d\d \d\.\d{12}E[+-]\d\d \d\.\d{12}E[+-]\d\d [01]",
                        first.split(b"\n")[130])
    assert first.count(b"\n") == 128 * 128
    env = dict(os.environ, LBM_WRITER_THREADS="1")
    r = subprocess.run([exe, *input_files("128x128")], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0 and open(tmp_path / "final_state.dat", "rb").read() == first
    # several slabs from the environment, same answer
    env = dict(os.environ, LBM_DEVICES="0,0")
    r = subprocess.run([exe, *input_files("128x128")], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert open(tmp_path / "final_state.dat", "rb").read() == first
    # LBM_NGPUS=N asks for devices 0..N-1: on a one-GPU box that must fail with the library's message, not crash
    env = dict(os.environ, LBM_NGPUS="2")
    env.pop("LBM_DEVICES", None)
    r = subprocess.run([exe, *input_files("128x128")], cwd=tmp_path, capture_output=True, text=True, env=env)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 1 and "device index 1 out of range" in r.stderr, r.stderr
    else:
        assert r.returncode == 0 and open(tmp_path / "final_state.dat", "rb").read() == first


# ---- full benchmark size: properties that need no full-size oracle run ------------------------------------

def test_x_tiling_invariance_8192x1024(lbm):
    """8 copies of obstacles_1024x1024 side by side (periodic in x, one accelerated row): av_vels must equal the
    1024x1024 golden record; exercises the 8192-wide rows of the benchmark grid against a reference file"""
    _, ob1 = lbm.read_inputs(*input_files("1024x1024"))
    ob = np.tile(ob1, (1, 8))
    nsteps = 3000
    p = lbm.make_params(8192, 1024, nsteps, 10, 0.1, 0.01, 1.85, ob)
    _, av = run_gpu(lbm, p, ob, None, nsteps)
    ref = golden_cols("1024x1024.av_vels.dat", [1])[:nsteps]
    assert np.max(np.abs(100.0 * (ref - av) / av)) < 0.1  # %


def test_8192x8192_vs_oracle_and_mass(lbm, oracle_f32_omp):
    nx = ny = 8192
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    nsteps = 6
    p = lbm.make_params(nx, ny, 64, obstacles=ob)
    po = oracle_params(oracle_f32_omp, p, ob)
    ref = oracle_f32_omp.init_cells(po)
    av_ref = oracle_f32_omp.run(po, ref, ob, nsteps)
    with lbm.LBM(p, ob) as sim:
        sim.upload(None)
        sim.run(nsteps)
        got, av = sim.download()
        assert max_rel(got, ref) < RTOL_CELLS and max_rel(av, av_ref) < RTOL_AV
        del ref
        m0 = got.astype(np.float64).sum()
        sim.run(58)
        got, av = sim.download()
    assert abs(got.astype(np.float64).sum() / m0 - 1.0) < 58 * 6e-8
    assert np.all(np.diff(av[:40]) > 0)  # the lid keeps accelerating the cavity from rest


@pytest.mark.parametrize("nsteps,fuse", [(20, -1), (16, 8)])
def test_benchmarked_configuration_vs_oracle(lbm, oracle_f32_omp, nsteps, fuse):
    """the configuration bench.py is quoted on, pinned on the oracle at the benchmarked depth: the 8192x8192 cavity with
    the library's default kernel for the driver's 20 timed steps (launches of 7 + 7 + 6 timesteps) and 16 steps as two
    depth-8 launches of the multi-round schedule (what the 400-step figure runs) — every distribution <= 2e-5, every
    av_vels entry <= 1e-4 relative against the fp32 oracle (kernels.cl:104-198 restated), not just against the
    single-step kernel"""
    nx = ny = 8192
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    po = oracle_params(oracle_f32_omp, p, ob)
    ref = oracle_f32_omp.init_cells(po)
    av_ref = oracle_f32_omp.run(po, ref, ob, nsteps)
    with lbm.LBM(p, ob) as sim:
        if fuse >= 0:
            sim.set_option("fuse", fuse)
        assert sim.get_option("fuse") == 8 and sim.get_option("launch_steps") == 8 and sim.get_option("multistep") == 0
        assert sim.get_option("pair") == 1   # (the default on one slab: chunk pairs, d2q9_deep_twin)
        assert sim.get_option("free_sweeps") == 1   # (several rounds of units: waves away from the walls sweep without obstacle handling)
        sim.upload(None)
        sim.run(nsteps)
        got, av = sim.download()
    assert max_rel(av, av_ref) < RTOL_AV
    # (row blocks: the relative error of 600M values without a second float64 copy of both states)
    worst = 0.0
    for y in range(0, ny, 512):
        worst = max(worst, max_rel(got[:, y:y + 512], ref[:, y:y + 512]))
    assert worst < RTOL_CELLS


def test_dead_neighbour_costs_one_timeout(lbm):
    """peer transport: a ring neighbour that never runs.  The first wait of the run gives up after halo_timeout_ms and
    raises the slab's error word; every later wait — 24 launch sets are queued behind it — must fall through at once
    (ONE timeout per run, not one per launch set), lbm_sync must report LBM_ERR_COMM and the context must refuse
    further work"""
    import time
    nx, ny, nsteps = 256, 64, 96     # two slabs of 32 rows; 24 launch sets at the halo depth of 4..8
    ob = np.zeros((ny, nx), np.int32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    a = lbm.LBM(p, ob, rank=0, nranks=2, device=0, comm=None)
    b = lbm.LBM(p, ob, rank=1, nranks=2, device=0, comm=None)
    try:
        ia, ib = a.peer_info(), b.peer_info()
        a.connect_peers(ib, ib)
        b.connect_peers(ia, ia)
        assert a.get_option("transport") == 3
        a.set_option("halo_timeout_ms", 300)
        assert a.get_option("halo_timeout_ms") == 300
        with pytest.raises(lbm.LBMError):
            a.set_option("halo_timeout_ms", 0)
        sets = -(-nsteps // a.get_option("launch_steps"))
        assert sets >= 12
        a.upload(None)
        t0 = time.perf_counter()
        a.run(nsteps)                 # rank 1 never runs: its flag words for rank 0 stay at zero
        with pytest.raises(lbm.LBMError, match="never came"):
            a.sync()
        el = time.perf_counter() - t0
        assert 0.25 < el < 0.3 * 4, "%d launch sets took %.2f s with a 0.3 s timeout" % (sets, el)
        with pytest.raises(lbm.LBMError):
            a.run(1)
    finally:
        a.close()
        b.close()


def test_obstacle_map_can_be_replaced(lbm, oracle_f32):
    """lbm_upload_obstacles (the reference's clEnqueueWriteBuffer(obstacles), d2q9-bgk.c:205-209, as an entry point of its
    own): a context whose map is replaced between runs continues exactly like a context created with the new map"""
    rng = np.random.default_rng(5)
    nx, ny, n = 256, 96, 12
    ob_a, cells0 = random_case(rng, nx, ny)
    ob_b = (rng.random((ny, nx)) < 0.08).astype(np.int32)
    p = lbm.make_params(nx, ny, 2 * n, obstacles=ob_b)
    p.free_cells_inv = np.float32(1.0)
    with lbm.LBM(p, ob_a) as sim:
        sim.upload(cells0)
        sim.run(n)
        mid, _ = sim.download()
        sim.upload_obstacles(ob_b)
        sim.run(n)
        got, av = sim.download()
    with lbm.LBM(p, ob_b) as sim:
        sim.upload(mid)
        sim.run(n)
        ref, av_ref = sim.download()
    assert np.array_equal(got, ref) and max_rel(av[n:], av_ref) < 2e-6
    po = oracle_params(oracle_f32, p, ob_b)
    po.free_cells_inv = np.float32(1.0)
    orc = mid.copy()
    oracle_f32.run(po, orc, ob_b, n)
    assert max_rel(got, orc) < RTOL_CELLS


def test_y_extension_invariance_16384x16384(lbm):
    """a grid whose arrays exceed 2^31 floats (16384x16384: 2.4e9 floats per grid): for fewer steps than rows the flow
    only knows the rows around the accelerated row ny-2, so the 96 rows around it — and the velocity sums — must equal
    those of a 16384x1024 grid bit for bit (same obstacles there); everything below stays at rest.  Catches 32-bit
    index arithmetic on rows near the end of the big arrays"""
    nx, big, small, nsteps, win = 16384, 16384, 1024, 30, 64
    rng = np.random.default_rng(99)
    band = (rng.random((2 * win, nx)) < 0.02).astype(np.int32)   # obstacles only in the compared window
    res = {}
    for ny in (small, big):
        ob = np.zeros((ny, nx), np.int32)
        ob[ny - win:, :] = band[:win]
        ob[:win, :] = band[win:]
        p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
        p.free_cells_inv = np.float32(1.0)   # raw sums of |j|/rho, comparable between the two grids
        with lbm.LBM(p, ob) as sim:
            del ob
            sim.upload(None)
            sim.run(nsteps)
            cells, av = sim.download()
        res[ny] = (np.concatenate([cells[:, ny - win:, :], cells[:, :win, :]], axis=1).copy(), av,
                   cells[:, win:ny - win, :].max(axis=(1, 2)), cells[:, win:ny - win, :].min(axis=(1, 2)))
        del cells
    assert np.array_equal(res[small][0], res[big][0])
    assert max_rel(res[big][1], res[small][1]) < 2e-6
    for ny in (small, big):   # outside the window nothing has moved: every plane is still uniform
        assert np.array_equal(res[ny][2], res[ny][3])


def test_calibration_kernels(lbm):
    """the two roofline denominators of bench.py: float4 copy bandwidth and packed-FMA issue rate, both plausible for an MI355X"""
    assert 2000.0 < lbm.copy_bandwidth_gbps(1 << 28, 4) < 8000.0
    assert 10.0 < lbm.valu_rate_tera(4) < 45.0


def test_abi_error_behaviour(lbm):
    """non-zero return codes + lbm_last_error() messages instead of the reference's print-and-exit (d2q9-bgk.c:858-866)"""
    import ctypes
    lib = lbm.load_library()
    ob = np.zeros((8, 8), np.int32)
    ctx = ctypes.c_void_p()
    bad = lbm.make_params(2, 8, 4)
    assert lib.lbm_create(ctypes.byref(ctx), ctypes.byref(bad), ob.ctypes.data, 1, None) == 1  # LBM_ERR_ARG
    assert b"at least 3x3" in lib.lbm_last_error() and not ctx.value
    p = lbm.make_params(8, 8, 4, obstacles=ob)
    assert lib.lbm_create(ctypes.byref(ctx), ctypes.byref(p), None, 1, None) == 1
    devs = (ctypes.c_int * 1)(99)
    assert lib.lbm_create(ctypes.byref(ctx), ctypes.byref(p), ob.ctypes.data, 1, devs) == 1
    assert b"out of range" in lib.lbm_last_error()
    two = (ctypes.c_int * 2)(0, 0)
    assert lib.lbm_create(ctypes.byref(ctx), ctypes.byref(p), ob.ctypes.data, 2, two) == 0  # 4 rows per slab: allowed
    lib.lbm_destroy(ctx)
    three = (ctypes.c_int * 3)(0, 0, 0)
    assert lib.lbm_create(ctypes.byref(ctx), ctypes.byref(p), ob.ctypes.data, 3, three) == 1  # 2 rows per slab: refused
    assert b"rows per slab" in lib.lbm_last_error()
    with lbm.LBM(p, ob) as sim:
        with pytest.raises(lbm.LBMError, match="unknown option"):
            sim.set_option("no_such_knob", 1)
        with pytest.raises(lbm.LBMError):
            sim.set_option("variant", 9)
        # the options of the deep window kernel are range-checked like the others
        for key, bad in (("fuse", -2), ("fuse", 9), ("twin_steps", 1), ("twin_steps", 9), ("obst_paths", 2), ("edge_aware", 2)):
            with pytest.raises(lbm.LBMError):
                sim.set_option(key, bad)
        for key, good in (("twin_steps", 0), ("twin_steps", 8), ("obst_paths", -1), ("edge_aware", -1), ("fuse", 7), ("fuse", 5), ("fuse", -1)):
            sim.set_option(key, good)
        sim.set_option("twin_steps", 0)
        with pytest.raises(lbm.LBMError, match="max_iters"):
            sim.run(5)
        sim.run(4)
        assert sim.steps_done == 4 and sim.row_range() == (0, 8)
        sim.upload(None)
        assert sim.steps_done == 0  # upload resets the step counter
        sim.run(0)
        _, av = sim.download(cells=False)
        assert av.size == 0


@pytest.mark.parametrize("nx,ny", [(8200, 300), (4100, 517), (5000, 333), (16384, 130), (260, 4099), (65536, 72)])
def test_odd_large_shapes_all_kernels_agree(lbm, nx, ny):
    """row lengths that are no multiple of a wave's 256 cells, row counts that no chunk size divides, very wide and
    very tall grids: the two-step kernel, the LDS multi-step kernel and four slabs agree with single steps bit for bit"""
    rng = np.random.default_rng(nx + ny)
    ob, cells0 = random_case(rng, nx, ny, blocked=0.03)
    nsteps = 7
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    base, av_base = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    for opts, kw in ((FUSED2, {}), (FUSED3, {}), (FUSED4, {}), ({"multistep": 4}, {}), (FUSED2, {"devices": [0, 0, 0, 0]}),
                     (FUSED3, {"devices": [0, 0, 0]}), (FUSED4, {"devices": [0, 0]}), ({}, {"devices": [0, 0]})):
        got, av = run_gpu(lbm, p, ob, cells0, nsteps, opts, **kw)
        assert np.array_equal(got, base), (opts, kw)
        assert max_rel(av, av_base) < 2e-6


@pytest.mark.parametrize("nx,ny,chunk", [(256, 8, 0), (256, 37, 5), (512, 64, 32), (1024, 50, 7), (2048, 16, 16), (260, 33, 4),
                                         (8192, 24, 8)])
@pytest.mark.parametrize("nsteps", [3, 4, 5, 10])
@pytest.mark.parametrize("windows,bufs,pair", [(1, 1, 1), (1, 1, 0), (-1, 0, -1)])
def test_three_steps_per_launch_equals_single_steps(lbm, nx, ny, chunk, nsteps, windows, bufs, pair):
    """d2q9_step3 (three timesteps per launch; the two windows of intermediate rows in LDS, one row-set of loads in flight: the
    one form round 4 kept; d2q9_step3p = chunk pairs sharing their start-up rows): bit-identical to single steps; step counts
    that are no multiple of three finish with the two-step / single-step kernels"""
    rng = np.random.default_rng(3 * nx + ny + nsteps)
    ob, cells0 = random_case(rng, nx, ny)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    got, av = run_gpu(lbm, p, ob, cells0, nsteps, {"multistep": 0, "fuse": 3, "windows": windows, "load_bufs": bufs,
                                                   "pair": pair, "chunk_rows": chunk})
    assert np.array_equal(got, single)
    assert max_rel(av, av_single) < 2e-6


@pytest.mark.parametrize("nx,ny,chunk", [(256, 32, 0), (256, 37, 5), (512, 64, 32), (1024, 50, 7), (260, 33, 4), (8192, 32, 8),
                                         (1024, 300, 128), (2048, 130, 0), (1000, 77, 9)])
@pytest.mark.parametrize("nsteps", [2, 3, 6, 7, 8, 13, 20, 23])
@pytest.mark.parametrize("depth,obst_paths,pair,nt", [(6, 0, 0, -1), (8, 1, 0, -1), (8, 0, 1, -1), (7, 1, 1, -1), (8, 1, -1, -1),
                                                      (8, 1, 0, 1), (7, 1, 0, 1), (6, 1, 0, 1), (8, 1, 1, 1), (8, 1, -1, 1)])
def test_deep_window_kernel_equals_single_steps(lbm, nx, ny, chunk, nsteps, depth, obst_paths, pair, nt):
    """d2q9_deep (up to eight timesteps per launch; lanes of two cells, explicit packed collision, four LDS windows + up to
    three register windows, x-shifted planes read back from LDS already shifted; obst_paths = 1: a second collision path
    without the bounce-back selects for waves that hold no blocked cell): bit-identical to single steps; a run's steps are
    split into as few launches as possible, of equal depth (20 = 7+7+6: the depth is a launch argument; option fuse = 6..8
    is its limit), a single left-over step goes to the single-step kernel.  pair = 1: d2q9_deep_twin, two waves per
    workgroup on the chunks 2p / 2p+1 of a strip that start at their common boundary and hand each other their first row
    of every level (five steps per launch by default, up to eight with option twin_steps: the levels whose window lives in
    registers then receive the twin's row through an LDS mailbox); -1 = where the library pairs by itself.
    nt = 1 (non-temporal stores, as on the grids this kernel is the default for): launches of exactly 6, 7 or 8 timesteps run
    the kernel instantiated for that depth — general form of the row loop while the levels start up, then the steady form
    whose level chain is straight-line code (option "steady", on by default)"""
    rng = np.random.default_rng(6 * nx + ny + nsteps)
    ob, cells0 = random_case(rng, nx, ny)
    if obst_paths:
        ob[ny // 3: 2 * ny // 3, :] = 0     # a band of rows without blocked cells: both paths run
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as sim:
        opts = {"multistep": 0, "fuse": depth, "chunk_rows": chunk, "obst_paths": obst_paths, "pair": pair, "nt_stores": nt}
        if nt == 1:
            assert sim.get_option("steady") == 1
        if pair == 1:
            opts["twin_steps"] = depth   # twins of 7 / 8 steps per launch: the register windows' first rows go through the mailbox
        for k, v in opts.items():
            sim.set_option(k, v)
        assert sim.get_option("fuse") == depth
        if pair >= 0:
            assert sim.get_option("pair") == pair
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()
    assert np.array_equal(got, single)
    assert max_rel(av, av_single) < 2e-6


def sparse_obstacles(rng, nx, ny, rows, cols, blocked=0.05):
    """blocked cells only in the first `rows` rows and in the first `cols` columns: chunks of rows above the band are free
    of them in every strip right of the columns, the periodic wrap brings the band back under the top chunks"""
    ob = np.zeros((ny, nx), np.int32)
    ob[:rows, :] = rng.random((rows, nx)) < blocked
    ob[:, :cols] = rng.random((ny, cols)) < blocked
    return ob


@pytest.mark.parametrize("nx,ny,chunk", [(2048, 260, 0), (2048, 260, 24), (1024, 400, 40), (4096, 130, 16)])
@pytest.mark.parametrize("nsteps", [6, 7, 8, 23])
@pytest.mark.parametrize("pair", [0, 1])
def test_deep_window_kernel_sweeps_without_obstacle_handling(lbm, nx, ny, chunk, nsteps, pair):
    """the kernels instantiated per depth (6, 7, 8 timesteps per launch): a wave whose rows hold no blocked cell inside its
    strip — looked up in a map of (strip, stored row) bits built from the obstacle map — runs deep_sweep<..., FREE>: no mask
    loads, no per-level ballot.  Obstacles in a band of rows and a band of columns only, so that free and looking waves sit
    side by side (and twins of different kinds share a workgroup); the accelerated row and the periodic wrap onto the band fall
    into free-looking chunks.  Bit-identical to single steps, with the option on (default) and off."""
    rng = np.random.default_rng(nx + 3 * ny + nsteps)
    ob = sparse_obstacles(rng, nx, ny, 20, 200)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    for free in (1, 0):
        with lbm.LBM(p, ob) as sim:
            # (by default only launches of several rounds of units, or slabs without any blocked cell, run free sweeps)
            opts = {"multistep": 0, "fuse": 8, "chunk_rows": chunk, "pair": pair, "nt_stores": 1, "free_sweeps": free}
            if pair == 1:
                opts["twin_steps"] = 8
            for k, v in opts.items():
                sim.set_option(k, v)
            assert sim.get_option("fuse") == 8 and sim.get_option("pair") == pair and sim.get_option("free_sweeps") == free
            sim.upload(cells0)
            sim.run(nsteps)
            got, av = sim.download()
        assert np.array_equal(got, single), free
        assert max_rel(av, av_single) < 2e-6


def walls(nx, ny, rng=None, extra=0):
    """side walls only (what a rank's slab of a cavity looks like), optionally a few blocked cells elsewhere"""
    ob = np.zeros((ny, nx), np.int32)
    ob[:, 0] = ob[:, -1] = 1
    if extra:
        ob[rng.integers(0, ny, extra), rng.integers(0, nx, extra)] = 1
    return ob


@pytest.mark.parametrize("nx,ny", [(2048, 260), (4096, 131), (1024, 400)])
@pytest.mark.parametrize("nsteps", [8, 23])
@pytest.mark.parametrize("pair,nt,balance", [(1, 1, -1), (0, 1, -1), (1, -1, -1), (0, -1, -1), (1, 1, 0)])
def test_one_round_schedules_balance_wall_strips(lbm, nx, ny, nsteps, pair, nt, balance):
    """a launch that is ONE round of units ends with its slowest waves — those of the strips that hold blocked cells in most
    rows, a cavity's two wall strips.  Such strips get a second (virtual) strip: two waves per chunk, each on half of its
    rows (of a chunk PAIR for the twins, whose halves must stay neighbours).  Side walls + a few blocked cells elsewhere:
    bit-identical to single steps, lone kernel and twins, per-depth and any-depth kernels, free sweeps on beside it"""
    rng = np.random.default_rng(nx + ny + nsteps)
    ob = walls(nx, ny, rng, extra=20)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as sim:
        opts = {"multistep": 0, "fuse": 8, "pair": pair, "nt_stores": nt, "balance": balance}
        if pair == 1:
            opts["twin_steps"] = 8
        for k, v in opts.items():
            sim.set_option(k, v)
        assert sim.get_option("fuse") == 8 and sim.get_option("pair") == pair
        assert sim.get_option("balance") == (2 if balance else 0)           # the two wall strips
        assert sim.get_option("free_sweeps") == (1 if balance else 0)       # balanced: the free waves are the critical path
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()
    assert np.array_equal(got, single)
    assert max_rel(av, av_single) < 2e-6


@pytest.mark.parametrize("pair,transport,balance", [(-1, "peer", -1), (0, "peer", -1), (-1, "copy", -1), (-1, "peer", 0), (0, "copy", 0)])
def test_row_slabs_balance_wall_strips(lbm, pair, transport, balance, halo_defaults):
    """the same for a rank's slab (compact launch sets with chunk pairs, with the lone kernel, and the two-stream sets):
    8192x1408 with side walls over two slabs — what every rank of a row-partitioned cavity runs"""
    halo_defaults(transport=transport)
    rng = np.random.default_rng(79)
    nx, ny, nsteps = 8192, 1408, 23
    ob = walls(nx, ny, rng, extra=50)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0, 0]) as sim:
        sim.set_option("pair", pair)
        sim.set_option("balance", balance)
        assert sim.get_option("fuse") == 8 and sim.get_option("halo_depth") == 8
        assert sim.get_option("balance") == (2 if balance else 0) and sim.get_option("free_sweeps") == (1 if balance else 0)
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


def test_free_sweeps_follow_a_replaced_obstacle_map(lbm):
    """lbm_upload_obstacles rebuilds the map the free sweeps look their rows up in: a context created on an EMPTY obstacle
    map (every wave free) that is then handed a map with blocked cells must run like a context created on that map"""
    rng = np.random.default_rng(123)
    nx, ny, nsteps = 2048, 260, 16
    ob = sparse_obstacles(rng, nx, ny, 30, 300)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, np.zeros_like(ob)) as sim:
        for k, v in {"multistep": 0, "fuse": 8, "pair": 0, "nt_stores": 1}.items():
            sim.set_option(k, v)
        assert sim.get_option("free_sweeps") == 1     # nothing blocked: on by default
        sim.upload_obstacles(ob)
        # a one-round launch with blocked cells: on only because the few strips that are blocked in most rows get balanced
        assert sim.get_option("balance") > 0 and sim.get_option("free_sweeps") == 1
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()
    assert np.array_equal(got, single)


@pytest.mark.parametrize("pair,transport,chunk", [(-1, "peer", 0), (0, "peer", 0), (-1, "copy", 0), (-1, "peer", 16), (0, "peer", 16)])
def test_row_slabs_free_sweeps(lbm, pair, transport, chunk, halo_defaults):
    """the same in slab mode (d2q9_deep<..., PUSH> / d2q9_deep_twin<..., PUSH> and the two-stream launch sets): the map is in
    stored-row coordinates, halo rows included; 8192x1408 over two slabs, blocked cells in a band of rows that straddles the
    slab boundary and in a band of columns.  chunk = 16: chunks that short make the interior several rounds of units (the
    tapered schedule the slabs of a 2-GPU run of 8192x8192 get), where the free sweeps are on without being asked for"""
    halo_defaults(transport=transport)
    rng = np.random.default_rng(78)
    nx, ny, nsteps = 8192, 1408, 23
    ob = np.zeros((ny, nx), np.int32)
    ob[690:720, :] = rng.random((30, nx)) < 0.02
    ob[:, 4000:4300] = rng.random((ny, 300)) < 0.02
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    one, av_one = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob, devices=[0, 0]) as sim:
        sim.set_option("pair", pair)
        if chunk:
            sim.set_option("chunk_rows", chunk)
        else:
            # one round of units and blocked cells: on by default only where the strips blocked in most rows are few and balanced
            assert sim.get_option("free_sweeps") == (1 if sim.get_option("balance") else 0)
            sim.set_option("free_sweeps", 1)
        assert sim.get_option("fuse") == 8 and sim.get_option("halo_depth") == 8 and sim.get_option("free_sweeps") == 1
        sim.upload(cells0)
        sim.run(nsteps)
        many, av_many = sim.download()
    assert np.array_equal(one, many)
    assert max_rel(av_many, av_one) < 2e-6


@pytest.mark.parametrize("nx,ny,chunk", [(256, 8, 0), (256, 37, 5), (512, 64, 32), (1024, 50, 7), (2048, 16, 16), (260, 33, 4),
                                         (8192, 24, 8), (1024, 300, 128)])
@pytest.mark.parametrize("nsteps", [4, 5, 6, 7, 13])
@pytest.mark.parametrize("pair", [1, 0])
def test_four_steps_per_launch_equals_single_steps(lbm, nx, ny, chunk, nsteps, pair):
    """d2q9_step4 (four timesteps per launch: two LDS windows + one register window) and d2q9_step4p (chunk pairs that
    hand each other their first row of every level instead of computing it twice): bit-identical to single steps;
    the steps left over after the last full launch go to the three-step / two-step / single-step kernels"""
    rng = np.random.default_rng(4 * nx + ny + nsteps)
    ob, cells0 = random_case(rng, nx, ny)
    p = lbm.make_params(nx, ny, nsteps, obstacles=ob)
    single, av_single = run_gpu(lbm, p, ob, cells0, nsteps, SINGLE)
    with lbm.LBM(p, ob) as sim:
        for k, v in {"multistep": 0, "fuse": 4, "pair": pair, "chunk_rows": chunk}.items():
            sim.set_option(k, v)
        assert sim.get_option("fuse") == 4 and sim.get_option("pair") == pair
        sim.upload(cells0)
        sim.run(nsteps)
        got, av = sim.download()
    assert np.array_equal(got, single)
    assert max_rel(av, av_single) < 2e-6

"""Pins the CPU oracle (oracle/d2q9_oracle.c) on the reference's own golden data before it is
trusted as the checker of the HIP path: the six shipped check/*.dat files, the Reynolds numbers of
README.md:78,88,98 and the pressures leaked by the reference's checker transcripts."""
import io
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_cols, golden_path, input_files, write_av_vels, write_final_state

PRINT_PRECISION_PCNT = 1e-8  # golden files carry 13 significant digits -> ~5e-11 relative


def max_pcnt(ref, sim):
    diff = ref - sim
    return float(np.max(np.abs(100.0 * diff / (ref - diff))))


@pytest.mark.parametrize("size", ["128x128", "128x256"])
def test_fp64_oracle_reproduces_shipped_golden_files(oracle_f64_omp, size):
    orc = oracle_f64_omp
    p, obst = orc.load(*input_files(size))
    cells = orc.init_cells(p)
    av = orc.run(p, cells, obst, p.max_iters)
    _, _, _, pressure = orc.final_fields(p, cells, obst)
    assert max_pcnt(golden_cols("%s.av_vels.dat" % size, [1]), av) < PRINT_PRECISION_PCNT
    ref_p = golden_cols("%s.final_state.dat" % size, [5]).reshape(p.ny, p.nx)
    assert max_pcnt(ref_p, pressure) < PRINT_PRECISION_PCNT
    # obstacle flags and coordinates of the golden final_state match the parsed obstacle file
    ref_o = golden_cols("%s.final_state.dat" % size, [6]).reshape(p.ny, p.nx)
    assert np.array_equal(ref_o.astype(np.int32), obst)


def test_fp64_oracle_velocity_columns_match_golden(oracle_f64_omp):
    """columns 3-5 of final_state.dat (u_x, u_y, u) — the reference's own writer leaves stale zeros in
    u_x/u_y (d2q9-bgk.c:778-779 shadowing); the golden files hold real values and so does the oracle"""
    orc = oracle_f64_omp
    p, obst = orc.load(*input_files("128x128"))
    cells = orc.init_cells(p)
    orc.run(p, cells, obst, p.max_iters)
    ux, uy, u, _ = orc.final_fields(p, cells, obst)
    ref = golden_cols("128x128.final_state.dat", [2, 3, 4])
    for col, got in zip(ref.T, (ux, uy, u)):
        assert np.max(np.abs(col.reshape(p.ny, p.nx) - got)) < 1e-12


def test_long_runs_recorded_by_make_golden():
    """256x256 (80 000 steps) and 1024x1024 (20 000 steps) take minutes on a CPU; make_golden.py ran them
    with the same oracle, asserted print-precision agreement with the shipped av_vels files and stored the
    final states.  Here: the stored scalars against the reference's published Reynolds numbers and the stored
    256x256 pressures against the values leaked by the reference's checker transcripts."""
    with open(os.path.join(GOLDEN, "generated", "oracle_f64_scalars.json")) as f:
        scal = json.load(f)
    with open(os.path.join(GOLDEN, "transcripts.json")) as f:
        tr = json.load(f)
    for size, re_ref in tr["reynolds"].items():
        assert abs(scal[size]["reynolds"] / re_ref - 1.0) < 5e-12, size
    d = np.load(os.path.join(GOLDEN, "generated", "256x256.final_state.npz"))
    for pin in tr["pressure_256x256"]:
        got = d["pressure"][pin["ii"], pin["jj"]]
        assert abs(got - pin["ref"]) < 2e-14, pin
    # first/last av_vels of the stored runs equal the shipped golden files' first/last lines
    for size in ("256x256", "1024x1024"):
        av = golden_cols("%s.av_vels.dat" % size, [1])
        assert abs(scal[size]["av_vels_first"] / av[0] - 1) < 1e-10
        assert abs(scal[size]["av_vels_last"] / av[-1] - 1) < 1e-10
    # mass is conserved by every step: total density == nx*ny*density to fp64 rounding
    for size, s in scal.items():
        nx, ny = (int(v) for v in size.split("x"))
        assert abs(s["total_density"] / (nx * ny * 0.1) - 1) < 1e-9


def test_fp32_oracle_passes_reference_checker(oracle_f32_omp, tmp_path):
    """the like-for-like partner of the GPU kernel (fp32, pairwise momenta) is inside the 1 % gate"""
    from check.check import run_check
    orc = oracle_f32_omp
    assert orc.lib.oracle_pairwise_momentum() == 1
    p, obst = orc.load(*input_files("128x128"))
    cells = orc.init_cells(p)
    av = orc.run(p, cells, obst, p.max_iters)
    fs, avf = str(tmp_path / "final_state.dat"), str(tmp_path / "av_vels.dat")
    orc.write_values(p, cells, obst, av, fs, avf)
    out = io.StringIO()
    code, avd, fsd = run_check(golden_path("128x128.av_vels.dat", tmp_path), golden_path("128x128.final_state.dat", tmp_path),
                               avf, fs, 1.0, out)
    assert code == 0, out.getvalue()
    assert abs(avd["max_diff_pcnt"]) < 0.2 and abs(fsd["max_diff_pcnt"]) < 0.2


def test_serial_cli_matches_golden_prefix(tmp_path):
    """oracle/d2q9-bgk-serial-f64 keeps the reference's command line and file formats"""
    exe = os.path.join(ROOT, "oracle", "d2q9-bgk-serial-f64")
    env = dict(os.environ, ORACLE_MAX_ITERS="60")
    r = subprocess.run([exe, *input_files("128x128")], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "==done==" in r.stdout and "Reynolds number:\t\t" in r.stdout and "Elapsed time:\t\t\t" in r.stdout
    lines = open(tmp_path / "av_vels.dat").read().splitlines()
    import re
    assert len(lines) == 60 and re.fullmatch(r"0:\t1\.0942691533\d\dE-05", lines[0])  # "%d:\t%.12E"
    ref = golden_cols("128x128.av_vels.dat", [1])[:60]
    got = np.loadtxt(tmp_path / "av_vels.dat", usecols=[1])
    assert max_pcnt(ref, got) < PRINT_PRECISION_PCNT
    first = open(tmp_path / "final_state.dat").readline().split()
    assert len(first) == 7 and first[:2] == ["0", "0"] and first[6] == "1"
    # usage / error conventions (d2q9-bgk.c:183-186, 868-874)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Usage: ")
    r = subprocess.run([exe, "/nonexistent.params", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "could not open input parameter file: /nonexistent.params" in r.stderr


def test_oracle_step_properties(oracle_f64_omp, oracle_f32):
    """structure checks on one step: mass conservation, rest state is a fixed point without forcing,
    bounce-back swaps opposite speeds, duplicate obstacle lines count once"""
    orc = oracle_f64_omp
    p, obst = orc.load(*input_files("128x128"))
    assert round(1.0 / p.free_cells_inv) == 15876  # 512 lines, 508 unique blocked cells
    cells = orc.init_cells(p)
    m0 = orc.total_density(p, cells)
    orc.run(p, cells, obst, 50)
    assert abs(orc.total_density(p, cells) / m0 - 1) < 1e-12
    # no acceleration -> the uniform rest state never changes and av_vels is exactly 0
    q = orc.make_params(64, 32, 1, 10, 0.1, 0.0, 1.85)
    ob = np.zeros((32, 64), dtype=np.int32)
    ob[5, 7] = 1
    orc.set_obstacles(q, ob)
    c0 = orc.init_cells(q)
    c1 = c0.copy()
    av = orc.run(q, c1, ob, 3)
    # (the reference's left-to-right momentum sums leave a rounding residue at rest — SURVEY F7)
    assert np.all(av < 1e-15) and np.max(np.abs(c1 - c0)) < 1e-16
    # ... and exactly 0 with the pairwise momentum differences of the fp32 build / the GPU kernel
    q32 = oracle_f32.make_params(64, 32, 1, 10, 0.1, 0.0, 1.85)
    oracle_f32.set_obstacles(q32, ob)
    c32 = oracle_f32.init_cells(q32)
    assert np.all(oracle_f32.run(q32, c32, ob, 3) == 0.0)
    # a single obstacle cell reflects what streams into it
    rng = np.random.default_rng(0)
    src = (0.01 + 0.1 * rng.random((9, 32, 64))).astype(np.float64)
    dst = np.zeros_like(src)
    orc.timestep(q, src, dst, ob)
    y, x = 5, 7
    g = [src[0, y, x], src[1, y, x - 1], src[2, y - 1, x], src[3, y, x + 1], src[4, y + 1, x],
         src[5, y - 1, x - 1], src[6, y - 1, x + 1], src[7, y + 1, x + 1], src[8, y + 1, x - 1]]
    opp = [0, 3, 4, 1, 2, 7, 8, 5, 6]
    for k in range(9):
        assert dst[opp[k], y, x] == g[k]

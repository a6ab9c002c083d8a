"""Process contract of the thin C host `d2q9-bgk` (reference d2q9-bgk.c:183-191, 466-492, 571-586,
868-880): usage, per-field parse errors, obstacle-file errors — all of which happen before any GPU
call, so this runs on a CPU-only box."""
import os
import subprocess

import pytest

from conftest import ROOT, input_files

EXE = os.path.join(ROOT, "d2q9-bgk")


def run(args, cwd):
    return subprocess.run([EXE, *args], cwd=cwd, capture_output=True, text=True)


def test_both_executable_names_exist():
    assert os.access(EXE, os.X_OK) and os.access(EXE + ".exe", os.X_OK)  # Makefile:3 vs README.md:14-21


@pytest.mark.parametrize("args", [[], ["only_one"], ["a", "b", "c"]])
def test_usage(tmp_path, args):
    r = run(args, tmp_path)
    assert r.returncode == 1
    assert r.stderr == "Usage: %s <paramfile> <obstaclefile>\n" % EXE


def test_missing_files(tmp_path):
    r = run(["nope.params", "nope.dat"], tmp_path)
    assert r.returncode == 1
    assert "Error at line " in r.stderr and "could not open input parameter file: nope.params" in r.stderr
    params, _ = input_files("128x128")
    r = run([params, "nope.dat"], tmp_path)
    assert r.returncode == 1 and "could not open input obstacles file: nope.dat" in r.stderr


FIELDS = ["nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega"]


@pytest.mark.parametrize("nfields", range(7))
def test_param_file_field_errors(tmp_path, nfields):
    good = ["128", "128", "10", "10", "0.1", "0.005", "1.85"]
    (tmp_path / "p.params").write_text("\n".join(good[:nfields] + ["oops"]) + "\n")
    r = run(["p.params", "whatever"], tmp_path)
    assert r.returncode == 1
    assert r.stderr.endswith("could not read param file: %s\n" % FIELDS[nfields])


@pytest.mark.parametrize("text,msg", [
    ("1 2\n", "expected 3 values per line in obstacle file"),
    ("1 2 x\n", "expected 3 values per line in obstacle file"),
    ("128 0 1\n", "obstacle x-coord out of range"),
    ("-1 0 1\n", "obstacle x-coord out of range"),
    ("0 128 1\n", "obstacle y-coord out of range"),
    ("0 0 2\n", "obstacle blocked value should be 1"),
])
def test_obstacle_file_errors(tmp_path, text, msg):
    params, _ = input_files("128x128")
    (tmp_path / "o.dat").write_text("0 0 1\n" + text)
    r = run([params, "o.dat"], tmp_path)
    assert r.returncode == 1
    assert r.stderr.startswith("Error at line ") and r.stderr.endswith(msg + "\n")


def test_fails_loudly_without_gpu(tmp_path):
    import ctypes
    n = ctypes.c_int()
    if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    r = run(list(input_files("128x128")), tmp_path)
    assert r.returncode == 1 and "LBM error during 'creating context'" in r.stderr
    assert not (tmp_path / "av_vels.dat").exists()

"""The Python-3 restatement of the reference's checker (check/check.py) against the reference's own
checker transcripts and contract (reference check/check.py:59-147, Makefile:26-27)."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_path

from check.check import get_diff_values, run_check


@pytest.fixture(scope="module")
def gold(tmp_path_factory):
    d = tmp_path_factory.mktemp("gold")
    return golden_path("128x128.av_vels.dat", d), golden_path("128x128.final_state.dat", d), d


def test_self_check_reproduces_reference_transcript(gold):
    """profiles/0initial/128x128/check.txt: golden vs golden"""
    av, fs, _ = gold
    out = io.StringIO()
    code, _, _ = run_check(av, fs, av, fs, 1.0, out)
    with open(os.path.join(GOLDEN, "transcripts.json")) as f:
        expect = json.load(f)["self_check_128x128"]
    assert code == 0
    assert out.getvalue().splitlines() == expect


def test_cli_flags_and_exit_codes(gold):
    av, fs, d = gold
    cmd = [sys.executable, os.path.join(ROOT, "check", "check.py"), "--ref-av-vels-file=" + av,
           "--ref-final-state-file=" + fs, "--av-vels-file=" + av, "--final-state-file=" + fs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("Both tests passed!")
    # a 2 % error in one av_vels entry fails at the default 1 % and passes at --tolerance 5
    lines = open(av).read().splitlines()
    idx, val = lines[1234].split(":\t")
    lines[1234] = "%s:\t%.12E" % (idx, float(val) * 1.02)
    bad = os.path.join(str(d), "bad_av.dat")
    open(bad, "w").write("\n".join(lines) + "\n")
    cmd[4] = "--av-vels-file=" + bad
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 1 and "av_vels failed check" in r.stdout and "final state failed check" not in r.stdout
    assert "Biggest difference (at step 1234)" in r.stdout
    r = subprocess.run(cmd + ["--tolerance", "5"], capture_output=True, text=True)
    assert r.returncode == 0
    # required arguments are enforced by argparse (exit code 2)
    r = subprocess.run(cmd[:3], capture_output=True, text=True)
    assert r.returncode == 2


def test_structural_mismatches(gold, tmp_path):
    av, fs, _ = gold
    short = tmp_path / "short_av.dat"
    short.write_text("\n".join(open(av).read().splitlines()[:100]) + "\n")
    out = io.StringIO()
    code, _, _ = run_check(av, fs, str(short), fs, 1.0, out)
    assert code == 1 and "Different number of steps in av_vels files" in out.getvalue()
    rows = open(fs).read().splitlines()
    rows[0], rows[1] = rows[1], rows[0]
    swapped = tmp_path / "swapped_fs.dat"
    swapped.write_text("\n".join(rows) + "\n")
    out = io.StringIO()
    code, _, _ = run_check(av, fs, av, str(swapped), 1.0, out)
    assert code == 1 and "Final state files coordinates were not the same" in out.getvalue()


def test_percentage_is_relative_to_the_simulated_value():
    """check/check.py:86-87: 100*diff/(ref-diff) == 100*(ref-sim)/sim"""
    d = get_diff_values(np.array([1.0, 2.0, 4.0]), np.array([1.0, 2.5, 4.0]))
    assert d["max_diff_step"] == 1 and d["max_diff"] == -0.5
    assert d["max_diff_pcnt"] == pytest.approx(-20.0)
    assert d["total"] == 0.5


def test_non_finite_values_fail(gold, tmp_path):
    av, fs, _ = gold
    lines = open(av).read().splitlines()
    lines[10] = "10:\tNAN"
    bad = tmp_path / "nan_av.dat"
    bad.write_text("\n".join(lines) + "\n")
    out = io.StringIO()
    code, _, _ = run_check(av, fs, str(bad), fs, 1.0, out)
    assert code == 1 and "av_vels failed check" in out.getvalue()

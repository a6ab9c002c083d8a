"""AddressSanitizer + UBSan builds of everything that runs on the CPU (`make asan`; SURVEY.md section 5, row "race
detection / sanitizers": the reference had only a dead -DDEBUG target, Makefile_old:30-31, and a commented-out
feenableexcept, d2q9-bgk.c:66,181).  The C host runs against tests/cpu/lbm_stub.c — the ABI of include/lbm.h without
numerics — so that its mmap scanner, the parallel obstacle parse and the threaded final_state formatter execute
under the sanitizers on a CPU-only box; the oracle's serial drivers run a short case under the same flags."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, input_files

ASAN_EXE = os.path.join(ROOT, "d2q9-bgk-asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module", autouse=True)
def asan_builds():
    subprocess.run(["make", "-C", ROOT, "asan"], check=True, stdout=subprocess.DEVNULL)


def run(args, cwd, **env):
    return subprocess.run([ASAN_EXE, *args], cwd=cwd, capture_output=True, text=True, env=dict(ENV, **env))


def clean(r):
    return "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr


def test_host_error_paths_under_sanitizers(tmp_path):
    """the parse-error exits of the reference's contract (d2q9-bgk.c:466-492,571-586) with the sanitizers watching"""
    params, obst = input_files("128x128")
    for args, msg in (([], "Usage:"), (["nope", "nope"], "could not open input parameter file"), ([params, "nope"], "could not open input obstacles file")):
        r = run(args, tmp_path)
        assert r.returncode == 1 and msg in r.stderr and clean(r), r.stderr
    for text, msg in (("1 2\n", "expected 3 values"), ("1 2 x\n", "expected 3 values"), ("128 0 1\n", "x-coord out of range"),
                      ("0 128 1\n", "y-coord out of range"), ("0 0 2\n", "blocked value should be 1"), ("", None)):
        (tmp_path / "o.dat").write_text("0 0 1\n" + text)
        for threads in ("1", "3"):
            r = run([params, "o.dat"], tmp_path, LBM_PARSER_THREADS=threads)
            assert r.returncode == 1 and clean(r), r.stderr     # the stub refuses to create a context at the latest
            assert (msg or "LBM error during 'creating context'") in r.stderr


def test_host_full_output_path_under_sanitizers(tmp_path):
    """LBM_STUB_RUN=1: the stub 'runs', the host formats and writes both files (multi-threaded sprintf into fixed-size line
    buffers, d2q9-bgk.c:835,848-851 formats) — line count, line format and thread-count independence"""
    params, obst = input_files("128x256")
    outs = []
    for threads in ("1", "5"):
        d = tmp_path / ("w" + threads)
        d.mkdir()
        r = run([params, obst], d, LBM_STUB_RUN="1", LBM_WRITER_THREADS=threads, LBM_MAX_ITERS="37")
        assert r.returncode == 0 and clean(r) and "==done==" in r.stdout, r.stderr
        outs.append((d / "final_state.dat").read_bytes())
        av = (d / "av_vels.dat").read_text().splitlines()
        assert len(av) == 37 and av[0].startswith("0:\t9.99999997") and av[36].startswith("36:\t")
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 128 * 256
    first = outs[0].split(b"\n")[300].split()
    assert len(first) == 7 and first[0] == b"44" and first[1] == b"2"


def tiled_obstacle_text(rng, nx, ny, n):
    xs, ys = rng.integers(0, nx, n), rng.integers(0, ny, n)
    return "".join("%d %d 1\n" % (x, y) for x, y in zip(xs, ys)), xs, ys


@pytest.mark.parametrize("layout", ["lines", "one_line", "ragged"])
def test_parallel_obstacle_scan_equals_serial(tmp_path, layout):
    """the parallel scan (token triples wherever the line breaks are, duplicates counted once, first bad record in file
    order decides) against the serial loop: same accepted files — seen through av_vels.dat being written with the same
    free-cell count — and the same error for a bad record planted in any thread's range"""
    rng = np.random.default_rng(5)
    nx, ny = 512, 384
    (tmp_path / "p.params").write_text("%d\n%d\n3\n10\n0.1\n0.005\n1.85\n" % (nx, ny))
    text, xs, ys = tiled_obstacle_text(rng, nx, ny, 30000)
    if layout == "one_line":
        text = text.replace("\n", " ")
    elif layout == "ragged":
        text = text.replace(" 1\n", "\n1\t", 20000).replace("\n", "\n\n", 500)
    (tmp_path / "o.dat").write_text(text)
    for threads in ("1", "2", "7", "16"):
        r = run(["p.params", "o.dat"], tmp_path, LBM_STUB_RUN="1", LBM_PARSER_THREADS=threads, LBM_NO_OUTPUT="1")
        assert r.returncode == 0 and clean(r), r.stderr
    # a bad record early, in the middle and at the very end: every thread count reports it like the serial scan
    recs = text.split("1", 1)
    lines = ("".join("%d %d 1\n" % (x, y) for x, y in zip(xs, ys))).splitlines()
    for pos, bad, msg in ((10, "%d 0 1" % nx, "obstacle x-coord out of range"), (15000, "0 -1 1", "obstacle y-coord out of range"),
                          (29999, "3 3 7", "obstacle blocked value should be 1"), (29999, "3 3", "expected 3 values per line"),
                          (20000, "5 x 1", "expected 3 values per line")):
        mod = list(lines)
        mod[pos] = bad
        # a later error of another kind must not win over the first one
        if pos < 29000:
            mod[29500] = "0 0 9"
        (tmp_path / "b.dat").write_text("\n".join(mod) + "\n")
        seen = set()
        for threads in ("1", "4", "16"):
            r = run(["p.params", "b.dat"], tmp_path, LBM_STUB_RUN="1", LBM_PARSER_THREADS=threads)
            assert r.returncode == 1 and clean(r), r.stderr
            seen.add(r.stderr.strip().splitlines()[-1])
        assert seen == {msg} or (len(seen) == 1 and msg in seen.pop()), (pos, bad, seen)


def test_parallel_scan_counts_duplicates_once(tmp_path):
    """free_cells is decremented once per unique cell (d2q9-bgk.c:583-585; every shipped file lists its corners twice):
    the Reynolds/av_vels scale the host hands to the library must not depend on the number of parser threads"""
    nx, ny = 256, 256
    (tmp_path / "p.params").write_text("%d\n%d\n2\n10\n0.1\n0.005\n1.85\n" % (nx, ny))
    lines = ["%d %d 1" % (x, y) for y in range(0, ny, 2) for x in range(nx)] * 3     # every cell three times: 98304 records
    (tmp_path / "o.dat").write_text("\n".join(lines) + "\n")
    outs = set()
    for threads in ("1", "8"):
        r = subprocess.run([ASAN_EXE, "p.params", "o.dat"], cwd=tmp_path, capture_output=True, text=True,
                           env=dict(ENV, LBM_STUB_RUN="1", LBM_PARSER_THREADS=threads, LBM_NO_OUTPUT="1", LBM_STUB_PRINT="1"))
        assert r.returncode == 0 and clean(r), r.stderr
        outs.add([ln for ln in r.stderr.splitlines() if ln.startswith("stub: free_cells_inv")][0])
    assert len(outs) == 1 and "%.9e" % (1.0 / (nx * ny // 2)) in outs.pop()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_oracle_serial_driver_under_sanitizers(tmp_path, prec):
    """the oracle (the checker of every parity test) itself under ASan/UBSan: 60 steps of the 128x256 input (periodic in
    y: the wrap-around indexing), files written, same av_vels as the optimised build"""
    files = list(input_files("128x256"))
    res = {}
    for tag, exe in (("asan", "d2q9-bgk-serial-%s-asan" % prec), ("opt", "d2q9-bgk-serial-%s" % prec)):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([os.path.join(ROOT, "oracle", exe)] + files, cwd=d, capture_output=True, text=True,
                           env=dict(ENV, ORACLE_MAX_ITERS="60"))
        assert r.returncode == 0 and clean(r), r.stderr
        res[tag] = (d / "av_vels.dat").read_text()
    assert res["asan"] == res["opt"] and res["asan"].count("\n") == 60

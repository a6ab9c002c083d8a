"""CPU-side test of the grouped ring exchange (csrc/halo_exchange.h) with a failing mock transport: whatever fails in
the middle, the transport group is closed again and the first error is reported (round 1 returned from inside an
open ncclGroupStart).  Compiled with g++ — no GPU, no HIP, no RCCL."""
import os
import subprocess

from conftest import ROOT


def test_ring_exchange_closes_its_group_on_every_failure(tmp_path):
    exe = str(tmp_path / "halo_exchange_test")
    src = os.path.join(ROOT, "tests", "cpu", "halo_exchange_test.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fsanitize=address,undefined", src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "halo_exchange_test: ok" in r.stdout, r.stdout + r.stderr

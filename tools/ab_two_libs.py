"""ad-hoc: A/B of two builds of the library in one run (LBM_LIB picks the .so), interleaved
usage: python tools/ab_two_libs.py tools/<script>.py"""
import os, subprocess, sys
script = sys.argv[1] if len(sys.argv) > 1 else "tools/ab_head.py"
for rnd in range(2):
    for lib in ("liblbm_hip_base.so", "liblbm_hip.so"):
        env = dict(os.environ, LBM_LIB=lib)
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            print(lib, ln)

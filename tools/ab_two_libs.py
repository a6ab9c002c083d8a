"""ad-hoc: A/B of two builds of the library in one run (LBM_LIB picks the .so), interleaved"""
import os, subprocess, sys
for rnd in range(2):
    for lib in ("liblbm_hip_base.so", "liblbm_hip.so"):
        env = dict(os.environ, LBM_LIB=lib)
        r = subprocess.run([sys.executable, "tools/ab_head.py"], env=env, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            print(lib, ln)

"""ad-hoc: A/B of several builds of the library in one run (LBM_LIB picks the .so), interleaved
usage: python tools/ab_two_libs.py tools/<script>.py [lib.so ...]   (default: liblbm_hip_base.so liblbm_hip.so)"""
import os, subprocess, sys
script = sys.argv[1] if len(sys.argv) > 1 else "tools/ab_head.py"
libs = sys.argv[2:] or ["liblbm_hip_base.so", "liblbm_hip.so"]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, LBM_LIB=lib)
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            print("%-22s" % lib, ln, flush=True)
        if r.returncode:
            print(lib, "FAILED", r.stderr[-2000:], flush=True)

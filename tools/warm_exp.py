"""tools/warm_exp.py none | copy | small N | same N — where the cold-start penalty of a short run comes from: 20 timed steps of the
8192x8192 cavity after 5 warm-up steps, preceded by nothing / 100 copies of 1 GiB / N steps of a 1024x1024 grid / N steps of the
same grid.  (The driver times 20 steps after 5: profiles/r02_cold_start.txt.)"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import lbm_amd
nx = ny = 8192
ob = np.zeros((ny, nx), np.int32); ob[0,:]=ob[-1,:]=1; ob[:,0]=ob[:,-1]=1
p = lbm_amd.make_params(nx, ny, 2000, obstacles=ob)
mode = sys.argv[1]
with lbm_amd.LBM(p, ob) as sim:
    sim.upload(None)
    if mode == "copy":
        lbm_amd.copy_bandwidth_gbps(1 << 30, 100)
    elif mode == "small":
        ob2 = np.zeros((1024, 1024), np.int32)
        p2 = lbm_amd.make_params(1024, 1024, 100000, obstacles=ob2)
        with lbm_amd.LBM(p2, ob2) as s2:
            s2.upload(None); s2.run(int(sys.argv[2])); s2.sync()
    elif mode == "same":
        sim.run(int(sys.argv[2])); sim.sync()
    sim.run(5); sim.sync()
    ms = sim.run_timed(20)
    print(mode, sys.argv[2:], "20 steps: %.1f us/step  %.0f GLUPS" % (ms / 20 * 1e3, nx * ny * 20 / ms / 1e6))

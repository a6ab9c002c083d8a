"""tools/d2h_time.py — where the read-back time of the output stage goes (8192x8192): lbm_final_state (four columns, 1.07 GB) and
lbm_download (nine planes, 2.4 GB) into a fresh and into an already touched host array.  Round 3: 53 GB/s into touched pageable memory,
19.6 GB/s into a fresh array (page faults of the destination, not the transfer): no pinned staging needed in the library."""
import sys, time, ctypes
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import lbm_amd
nx = ny = 8192
ob = np.zeros((ny, nx), np.int32); ob[0,:]=ob[-1,:]=1; ob[:,0]=ob[:,-1]=1
p = lbm_amd.make_params(nx, ny, 64, obstacles=ob)
with lbm_amd.LBM(p, ob) as sim:
    sim.upload(None); sim.run(16); sim.sync()
    for touched in (False, True, True):
        outs = [np.empty((ny, nx), np.float32) for _ in range(4)]
        if touched:
            for o in outs: o.fill(0)
        t0 = time.perf_counter()
        lbm_amd._check(sim.lib.lbm_final_state(sim.ctx, *[o.ctypes.data for o in outs]), "fs")
        t = time.perf_counter() - t0
        print("final_state 4 x %.0f MB, destination %s: %.4f s = %.1f GB/s" % (nx*ny*4/1e6, "touched" if touched else "fresh", t, 4*nx*ny*4/t/1e9), flush=True)
    cells = np.empty((9, ny, nx), np.float32); cells.fill(0)
    t0 = time.perf_counter()
    lbm_amd._check(sim.lib.lbm_download(sim.ctx, cells.ctypes.data, None), "dl")
    t = time.perf_counter() - t0
    print("download 9 planes %.2f GB (touched): %.4f s = %.1f GB/s" % (cells.nbytes/1e9, t, cells.nbytes/t/1e9))

"""PCIe-inclusive rate of the C-ABI boundary when the caller hands over host arrays (lbm_upload / lbm_download of the
reference's float[9][ny][nx]) instead of letting the library build the initial state on the device."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
out = av = cells = None
for (nx, ny, steps) in [(8192, 8192, 999), (1024, 1024, 19998)]:
    out = av = cells = None  # free the previous case's arrays outside the timed regions
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, steps, obstacles=ob)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float32).reshape(9, 1, 1) * np.float32(0.1)
    cells = np.ascontiguousarray(np.broadcast_to(w, (9, ny, nx))).astype(np.float32)
    with lbm_amd.LBM(p, ob) as sim:
        sim.upload(cells); sim.sync()      # first touch of the staging path
        t0 = time.perf_counter(); sim.upload(cells); sim.sync(); t_up = time.perf_counter() - t0
        t0 = time.perf_counter(); sim.run(steps); sim.sync(); t_loop = time.perf_counter() - t0
        t0 = time.perf_counter(); out, av = sim.download(); t_down = time.perf_counter() - t0
    gb = cells.nbytes / 1e9
    print("%dx%d %d steps: upload %.3f s (%.1f GB/s), loop %.3f s (%.0f MLUPS), download %.3f s (%.1f GB/s); "
          "PCIe-inclusive %.0f MLUPS" % (nx, ny, steps, t_up, gb / t_up, t_loop, nx * ny * steps / t_loop / 1e6, t_down, gb / t_down,
                                         nx * ny * steps / (t_up + t_loop + t_down) / 1e6), flush=True)

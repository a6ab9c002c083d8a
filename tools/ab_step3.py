"""ad-hoc: d2q9_step3 (LDS windows, one load buffer) and d2q9_step2 on a few sizes + a checksum of the grid"""
import sys, zlib
import numpy as np
sys.path.insert(0, '.')
import lbm_amd
for (nx, ny, steps) in [(2048, 2048, 960), (8192, 8192, 240), (8192, 1024, 480), (1024, 1024, 1920)]:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    if ny == 2048: ob[700:900, 500:600] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0)
        for (fuse, chunk, cmin) in [(3, 12, 4), (3, 16, 4), (1, 0, 0)]:
            sim.set_option("fuse", fuse); sim.set_option("windows", 1); sim.set_option("load_bufs", 1)
            sim.set_option("chunk_min", cmin); sim.set_option("chunk_rows", chunk)
            sim.upload(None); sim.run(25)
            crc = ""
            if ny == 2048:
                cells, av = sim.download()
                crc = "crc %08x" % zlib.crc32(cells.tobytes())
            best = min(sim.run_timed(steps) for _ in range(3))
            print("%5dx%-5d fuse=%d chunk=%-2d min=%-2d us/step %9.3f MLUPS %8.0f %s" % (nx, ny, fuse, chunk, cmin, best / steps * 1e3, nx * ny * steps / best / 1e3, crc), flush=True)

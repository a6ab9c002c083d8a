"""Same-box A/B of d2q9_step3: register windows (1 wave/SIMD) vs LDS windows (2 waves/SIMD), 1 or 2 load buffers,
over a few chunk schedules.  Also checks that every variant gives the same grid bit for bit."""
import sys
import numpy as np
sys.path.insert(0, '.')
import lbm_amd

sizes = [(8192, 8192, 240), (8192, 1024, 480), (2048, 2048, 960)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (nx, ny, steps) in sizes:
    ob = np.zeros((ny, nx), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, 100000, obstacles=ob)
    ref = None
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("multistep", 0); sim.set_option("fuse", 3)
        for rnd in range(2):
            for (win, bufs, chunk, cmin) in [(0, 2, 0, 0), (1, 1, 0, 0), (1, 1, 32, 6), (1, 1, 24, 6), (1, 1, 16, 4), (1, 1, 12, 4), (1, 2, 32, 6), (0, 1, 0, 0)]:
                sim.set_option("windows", win); sim.set_option("load_bufs", bufs)
                sim.set_option("chunk_min", cmin); sim.set_option("chunk_rows", chunk)
                sim.upload(None); sim.run(24)
                if rnd == 0 and ny <= 2048:
                    cells, _ = sim.download()
                    if ref is None: ref = cells
                    else: assert np.array_equal(ref, cells), "variant differs"
                best = min(sim.run_timed(steps) for _ in range(2))
                print("%5dx%-5d windows=%d bufs=%d chunk=%-2d min=%-2d us/step %9.3f MLUPS %8.0f" % (nx, ny, win, bufs, chunk, cmin, best / steps * 1e3, nx * ny * steps / best / 1e3), flush=True)

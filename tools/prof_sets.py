#!/usr/bin/env python3
"""tools/prof_sets.py nx ny [key=value ...] — lbm_run_profiled of a ring-of-one slab (peer transport) and of the same
grid as a plain single slab: mean microseconds of the edge launch / exchange / interior launch and the launch-set
period, i.e. where a launch set's time goes (the same numbers bench.py --gpus N prints per rank)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd
nx, ny = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:])
ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
for label, ring, extra in (("single slab, no halo rows", False, {}), ("ring of one, compact", True, {}), ("ring of one, two streams", True, {"compact": 0})):
    lbm_amd.set_default("force_halo", 1 if ring else 0)
    lbm_amd.set_default("transport", "peer" if ring else "auto")
    p = lbm_amd.make_params(nx, ny, 4096, obstacles=ob)
    with lbm_amd.LBM(p, ob, **(dict(devices=[0]) if ring else {})) as sim:
        for k, v in dict(opts, **extra).items():
            if ring or k != "compact":
                sim.set_option(k, int(v))
        sim.upload(None); sim.run(64)
        per = max(sim.get_option("multistep"), {0: 1, 1: 2, 3: 3, 4: 4, 6: 6, 7: 7, 8: 8}[sim.get_option("fuse")])
        st = sim.run_profiled(32 * per)
        ms = sim.run_timed(64 * per)
        print("%-26s %s  unprofiled %.2f us/step = %.0f MLUPS" % (label, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()},
              ms / (64 * per) * 1e3, nx * ny * 64 * per / ms / 1e3), flush=True)

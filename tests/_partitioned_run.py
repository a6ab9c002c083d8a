"""Helper of test_gpu_parity.py::test_partitioned_full_run_passes_checker — one full-length run of a shipped input
in a row-partitioned configuration, in a subprocess (a hang in a transport is then a test failure, not a stuck
session).  Writes av_vels.dat / final_state.dat in the reference's format into <outdir> plus state.npz (the four
output columns and av_vels as arrays, for the bit-identity check against the undivided run).
modes: single | slabs8 (8 slabs on device 0, peer stores) | slabs8_copy | ring_rccl | ring_peer (one rank that is
its own ring neighbour: the one-process-per-GPU code path with the RCCL / peer transport)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd
from conftest import input_files, write_av_vels, write_final_state

mode, size, outdir = sys.argv[1:4]
p, obst = lbm_amd.read_inputs(*input_files(size))
kw = {}
if mode.startswith("slabs8"):
    lbm_amd.set_default("transport", "copy" if mode.endswith("copy") else "peer")
    kw = dict(devices=[0] * 8)
elif mode == "ring_rccl":
    lbm_amd.set_default("force_halo", 1)
    lbm_amd.set_default("transport", "rccl")
    kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id())
elif mode == "ring_peer":
    lbm_amd.set_default("force_halo", 1)
    lbm_amd.set_default("transport", "peer")
    kw = dict(devices=[0])
with lbm_amd.LBM(p, obst, **kw) as sim:
    if mode != "single":
        assert sim.get_option("transport") == {"slabs8": 3, "slabs8_copy": 2, "ring_rccl": 1, "ring_peer": 3}[mode]
        assert sim.get_option("halo_depth") >= 2
    sim.upload(None)
    sim.run(p.max_iters)
    _, av = sim.download(cells=False)
    fields = sim.final_state()
    re = sim.reynolds()
    print("mode %s: %d slabs, transport %d, multistep %d, fuse %d, Re %.9e" % (
        mode, sim.get_option("nslabs"), sim.get_option("transport") if mode != "single" else 0,
        sim.get_option("multistep"), sim.get_option("fuse"), re))
write_final_state(os.path.join(outdir, "final_state.dat"), obst, *fields)
write_av_vels(os.path.join(outdir, "av_vels.dat"), av)
np.savez(os.path.join(outdir, "state.npz"), av=av, ux=fields[0], uy=fields[1], u=fields[2], pr=fields[3], re=re)
print("partitioned run ok")

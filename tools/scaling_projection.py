"""tools/scaling_projection.py — what one rank of an N-GPU run does, measured on ONE GPU: a slab of the size an
N-way row partition gives each rank exchanges halos with itself (default force_halo, ring of one; transport = argv[1]: peer | rccl | copy), so the
edge launch / ncclSend+ncclRecv / interior launch structure and its overlap are the real ones; only the peer is the
same GPU.  Prints the per-rank rate and N x that rate as the PROJECTED whole-job rate (no xGMI latency in it)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd
TRANSPORT = sys.argv[1] if len(sys.argv) > 1 else "peer"   # rccl | peer | copy


def rate(nx, rows, steps, ring):
    ob = np.zeros((rows, nx), np.int32); ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, rows, steps * 4 + 64, obstacles=ob)
    kw = {}
    if ring:
        kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) if TRANSPORT == "rccl" else dict(devices=[0])
    lbm_amd.set_default("force_halo", 1 if ring else 0)
    lbm_amd.set_default("transport", TRANSPORT if ring else "auto")
    with lbm_amd.LBM(p, ob, **kw) as sim:
        sim.upload(None); sim.run(48)
        ms = min(sim.run_timed(steps) for _ in range(3))
        f = sim.get_option("fuse")
        kern = "resident x%d" % sim.get_option("launch_steps") if sim.get_option("resident") else "multi x%d" % sim.get_option("multistep") if sim.get_option("multistep") else (
            "deep%s x%d" % ("_twin" if sim.get_option("pair") else "", sim.get_option("launch_steps")) if f >= 5 else {0: "step", 1: "step2", 3: "step3", 4: "step4"}[f])
    return ms / steps * 1e3, kern


print("transport: %s" % TRANSPORT, flush=True)
for (nx, ny, steps1) in [(8192, 8192, 240), (1024, 1024, 3840)]:
    us1, k1 = rate(nx, ny, steps1, False)
    base = nx * ny / us1
    print("%dx%d  1 GPU (no halos)            %9.2f us/step %9.0f MLUPS  %-9s" % (nx, ny, us1, base, k1), flush=True)
    for n in (2, 4, 8):
        rows = ny // n
        us, k = rate(nx, rows, steps1 * min(n, 4), True)
        print("%dx%d  %d GPUs: rank slab %5dx%-5d %9.2f us/step %9.0f MLUPS per rank  %-9s projected %9.0f MLUPS = %.2fx" % (
            nx, ny, n, nx, rows, us, nx * rows / us, k, n * nx * rows / us, n * nx * rows / us / base), flush=True)

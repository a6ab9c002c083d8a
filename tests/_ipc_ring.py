"""Helper of test_gpu_parity.py::test_peer_transport_between_processes — WORLD ranks as separate PROCESSES on the one
GPU of the box (torch.distributed over gloo carries the descriptors and adds up the velocity records; RCCL is not
involved: it refuses two ranks on one device).  Every rank owns a row slab, maps its ring neighbours' grids and
flag words with hipIpcOpenMemHandle, pushes its edge rows straight into them and waits on its own flag words:
the one-process-per-GPU data path of the peer transport with real process boundaries.  Rank 0 compares the
gathered state with an undivided single-slab run, bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    import lbm_amd

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nx, ny, nsteps, fuse, multistep, sync = (int(v) for v in sys.argv[1:7])
    walls = len(sys.argv) > 7 and sys.argv[7] == "walls"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(33)
    ob = (rng.random((ny, nx)) < (0.00001 if walls else 0.04)).astype(np.int32)
    ob[0, :] = ob[-1, :] = 0
    if walls:       # a cavity's side walls: every rank's one-round launch sets balance the two wall strips and run free sweeps
        ob[:, 0] = ob[:, -1] = 1
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)

    sim = lbm_amd.LBM(p, ob, rank=rank, nranks=world, device=0, comm=None)
    assert sim.get_option("transport") == 0          # nothing yet: lbm_run must refuse
    try:
        sim.run(1)
        raise AssertionError("lbm_run worked without a transport")
    except lbm_amd.LBMError as e:
        assert "transport" in str(e)
    infos = [None] * world
    dist.all_gather_object(infos, sim.peer_info())
    sim.connect_peers(infos[(rank - 1) % world], infos[(rank + 1) % world])
    assert sim.get_option("transport") == 3
    sim.set_option("halo_sync", sync)
    sim.set_option("fuse", fuse)
    sim.set_option("multistep", multistep)
    if walls:
        assert sim.get_option("balance") == 2 and sim.get_option("free_sweeps") == 1
    sim.upload(cells0)
    dist.barrier()
    # split runs: the exchange counter runs on across lbm_run calls, ranks drift apart in between
    sim.run(5)
    if rank % 2:
        sim.sync()
    sim.run(nsteps - 5)
    got, av = sim.download()          # own rows of the global array; velocity sums over own rows
    y0, y1 = sim.row_range()
    part = torch.zeros((9, ny, nx), dtype=torch.float32)
    part[:, y0:y1, :] = torch.from_numpy(got[:, y0:y1, :])
    dist.all_reduce(part)             # rows are disjoint: the sum is the assembled state
    avt = torch.from_numpy(av.astype(np.float64))
    dist.all_reduce(avt)
    dist.barrier()
    sim.disconnect_peers()            # tear-down between processes: unmap, meet, and only then free (see lbm.h)
    dist.barrier()
    sim.close()
    if rank == 0:
        with lbm_amd.LBM(p, ob) as one:
            one.set_option("fuse", 0)
            one.set_option("multistep", 0)
            one.upload(cells0)
            one.run(nsteps)
            ref, av_ref = one.download()
        if not np.array_equal(part.numpy(), ref):
            bad = np.argwhere(np.any(part.numpy() != ref, axis=(0, 2))).ravel()
            raise AssertionError("assembled state differs from the single-slab run in %d rows: %s" % (bad.size, bad))
        assert np.max(np.abs(avt.numpy() - av_ref) / av_ref) < 2e-6
        print("ipc-ring ok: %d processes, %dx%d, %d steps, fuse %d multistep %d sync %d" % (world, nx, ny, nsteps, fuse, multistep, sync))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

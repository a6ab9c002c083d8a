"""World-size 2-4 `gloo` rehearsal of the row partition on CPU: every rank steps its slab with the oracle's
row kernels, exchanges halo rows with its ring neighbours every `depth` steps (recomputing a shrinking
halo region in between) and all-reduces the velocity sums — the protocol liblbm_hip.so runs per GPU with
RCCL (csrc/lbm_hip.cpp: run_steps / exchange_halos) for halo depths 1, 2 (two-step kernel), 3, 4, 5 (the five-step
chunk pairs of mid-size slabs) and 8 (d2q9_multi, the deep window kernel).  The result must equal the undivided oracle run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, input_files


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, size, nsteps, depth, out_dir):
    """One rank of the row partition with halo depth `depth`: the slab stores `depth` halo rows below and above its
    own rows, exchanges them every `depth` steps and in between recomputes a halo region that shrinks by one row per
    step — depth 1 is the classic per-step exchange, 2 what the two-step kernel needs, 8 what d2q9_multi uses."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import lbm_amd
    from oracle.oracle import Oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle("f32")
    pg, obst = orc.load(*input_files(size))
    nx, ny = pg.nx, pg.ny
    y0, rows = lbm_amd.slab_rows(ny, world, rank)
    south, north = lbm_amd.ring_neighbours(world, rank)
    H = depth
    L = rows + 2 * H                      # stored rows: stored row e <-> global row (y0 - H + e) mod ny
    gy = [(y0 - H + e) % ny for e in range(L)]
    p = orc.make_params(nx, L, nsteps, pg.reynolds_dim, pg.density, pg.accel, pg.omega)
    ob = np.ascontiguousarray(obst[gy])   # halo rows carry the neighbours' obstacle flags
    cur = np.zeros((9, L, nx), dtype=np.float32)
    cur[:, H:H + rows] = orc.init_cells(pg)[:, y0:y0 + rows]
    nxt = np.zeros_like(cur)
    sums = np.zeros(nsteps, dtype=np.float64)

    def exchange(grid):
        # my top `depth` rows feed the north neighbour's south halo, my bottom rows the south neighbour's north halo
        # (whole rows, as exchange_halos in csrc/lbm_hip.cpp sends them; lbm_amd.HALO_PLANES names the planes that
        # actually cross the edge in a single step)
        up = torch.from_numpy(np.ascontiguousarray(grid[:, rows:rows + H]))
        down = torch.from_numpy(np.ascontiguousarray(grid[:, H:2 * H]))
        from_s, from_n = torch.empty_like(up), torch.empty_like(down)
        reqs = [dist.isend(up, north, tag=1), dist.isend(down, south, tag=2),
                dist.irecv(from_s, south, tag=1), dist.irecv(from_n, north, tag=2)]
        for r in reqs:
            r.wait()
        grid[:, 0:H] = from_s.numpy()
        grid[:, H + rows:] = from_n.numpy()

    t = 0
    while t < nsteps:
        exchange(cur)
        for s in range(1, min(H, nsteps - t) + 1):
            # accelerate_flow acts on EVERY stored copy of global row ny-2 that is still inside the valid region
            for e in range(s - 1, L - s + 1):
                if gy[e] == ny - 2:
                    orc.accelerate_row(p, cur, ob, e)
            # the valid region shrinks by one row per step; only the slab's own rows count for av_vels
            orc.timestep_rows(p, cur, nxt, ob, s, H)
            sums[t] = orc.timestep_rows(p, cur, nxt, ob, H, H + rows)
            orc.timestep_rows(p, cur, nxt, ob, H + rows, L - s)
            cur, nxt = nxt, cur
            t += 1
    tot = torch.from_numpy(sums.copy())
    dist.all_reduce(tot)  # the per-rank partial velocity sums (ncclAllReduce in the library)
    av = (tot.numpy() * float(pg.free_cells_inv)).astype(np.float32)
    np.save(os.path.join(out_dir, "cells_%d.npy" % rank), cur[:, H:H + rows])
    if rank == 0:
        np.save(os.path.join(out_dir, "av.npy"), av)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size,nsteps,depth", [(2, "128x128", 40, 1), (3, "128x256", 25, 1), (2, "128x128", 21, 2),
                                                     (2, "128x256", 27, 8), (4, "128x256", 16, 8), (3, "128x256", 22, 4), (2, "128x128", 14, 3),
                                                     # five halo rows per launch set of five steps: the slab pairs of round 4 (23 = 5+5+5+5+3)
                                                     (2, "128x256", 23, 5), (4, "128x256", 17, 5)])
def test_row_partition_protocol_matches_undivided_run(tmp_path, world, size, nsteps, depth):
    import torch.multiprocessing as mp
    import lbm_amd
    from oracle.oracle import Oracle

    mp.spawn(_rank_main, args=(world, _free_port(), size, nsteps, depth, str(tmp_path)), nprocs=world, join=True)
    orc = Oracle("f32")
    p, obst = orc.load(*input_files(size))
    ref = orc.init_cells(p)
    av_ref = orc.run(p, ref, obst, nsteps)
    got = np.concatenate([np.load(tmp_path / ("cells_%d.npy" % r)) for r in range(world)], axis=1)
    assert np.array_equal(got, ref)  # same arithmetic per cell -> bit-identical state
    av = np.load(tmp_path / "av.npy")
    assert np.max(np.abs(av - av_ref) / av_ref) < 1e-6  # fp64 partial sums, different grouping


def test_slab_geometry_helpers():
    import lbm_amd
    for ny, P in [(128, 2), (1024, 8), (8192, 8), (50, 3), (257, 8)]:
        spans = [lbm_amd.slab_rows(ny, P, r) for r in range(P)]
        assert spans[0][0] == 0 and sum(r for _, r in spans) == ny
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(P - 1))
        assert max(r for _, r in spans) - min(r for _, r in spans) <= 1
        owners = [r for r in range(P) if lbm_amd.accel_row_local(ny, *spans[r]) >= 0]
        assert len(owners) == 1  # exactly one slab owns the accelerated row ny-2
        y0, rows = spans[owners[0]]
        assert y0 + lbm_amd.accel_row_local(ny, y0, rows) == ny - 2
    assert lbm_amd.ring_neighbours(8, 0) == (7, 1) and lbm_amd.ring_neighbours(8, 7) == (6, 0)
    assert lbm_amd.ring_neighbours(1, 0) == (0, 0)


def _bench_plumbing(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import bench
    dist = bench.init_dist("gloo", rank, world)
    blob = bytes(range(128)) if rank == 0 else None
    got = bench.share_comm_id(dist, rank, blob, 128, torch.device("cpu"))
    mx = bench.max_over_ranks(dist, [1.0 + rank, 10.0 - rank], torch.device("cpu"))
    dist.barrier()
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write("%s %s" % (got == bytes(range(128)), mx))
    dist.destroy_process_group()


def test_bench_distributed_plumbing(tmp_path):
    """bench.py's rank plumbing (process group, RCCL-id broadcast, max-over-ranks timing) with gloo, 2 ranks"""
    import torch.multiprocessing as mp
    mp.spawn(_bench_plumbing, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / ("r%d.txt" % r)).read() == "True [2.0, 10.0]"

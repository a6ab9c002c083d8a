"""World-size-2 (and 3) `gloo` rehearsal of the row partition on CPU: every rank steps its slab with the
oracle's row kernels, exchanges the three distributions that cross each slab edge with its ring
neighbours and all-reduces the velocity sums — exactly the protocol liblbm_hip.so runs per GPU with
RCCL (csrc/lbm_hip.cpp: run_steps / exchange_halos).  The result must equal the undivided oracle run."""
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


def _rank_main(rank, world, port, size, nsteps, out_dir):
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
    accel_row = lbm_amd.accel_row_local(ny, y0, rows)
    # local grid with one halo row below and above: local row i <-> global row y0 + i - 1
    L = rows + 2
    p = orc.make_params(nx, L, nsteps, pg.reynolds_dim, pg.density, pg.accel, pg.omega)
    ob = np.zeros((L, nx), dtype=np.int32)
    ob[1:-1] = obst[y0:y0 + rows]
    cur = np.zeros((9, L, nx), dtype=np.float32)
    cur[:, 1:-1] = orc.init_cells(pg)[:, y0:y0 + rows]
    nxt = np.zeros_like(cur)
    sums = np.zeros(nsteps, dtype=np.float64)

    def exchange(grid):
        # planes 2,5,6 of my top row feed the north neighbour's south halo; 4,7,8 of my bottom row feed the
        # south neighbour's north halo (lbm_amd.HALO_PLANES mirrors exchange_halos in csrc/lbm_hip.cpp)
        up = torch.from_numpy(np.ascontiguousarray(grid[lbm_amd.HALO_PLANES["to_north"], rows]))
        down = torch.from_numpy(np.ascontiguousarray(grid[lbm_amd.HALO_PLANES["to_south"], 1]))
        from_s, from_n = torch.empty_like(up), torch.empty_like(down)
        reqs = [dist.isend(up, north, tag=1), dist.isend(down, south, tag=2),
                dist.irecv(from_s, south, tag=1), dist.irecv(from_n, north, tag=2)]
        for r in reqs:
            r.wait()
        grid[lbm_amd.HALO_PLANES["to_north"], 0] = from_s.numpy()
        grid[lbm_amd.HALO_PLANES["to_south"], rows + 1] = from_n.numpy()

    for t in range(nsteps):
        if accel_row >= 0:
            orc.accelerate_row(p, cur, ob, accel_row + 1)
        exchange(cur)
        sums[t] = orc.timestep_rows(p, cur, nxt, ob, 1, rows + 1)
        cur, nxt = nxt, cur
    tot = torch.from_numpy(sums.copy())
    dist.all_reduce(tot)  # the per-rank partial velocity sums (ncclAllReduce in the library)
    av = (tot.numpy() * float(pg.free_cells_inv)).astype(np.float32)
    np.save(os.path.join(out_dir, "cells_%d.npy" % rank), cur[:, 1:-1])
    if rank == 0:
        np.save(os.path.join(out_dir, "av.npy"), av)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size,nsteps", [(2, "128x128", 40), (3, "128x256", 25)])
def test_row_partition_protocol_matches_undivided_run(tmp_path, world, size, nsteps):
    import torch.multiprocessing as mp
    import lbm_amd
    from oracle.oracle import Oracle

    mp.spawn(_rank_main, args=(world, _free_port(), size, nsteps, str(tmp_path)), nprocs=world, join=True)
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

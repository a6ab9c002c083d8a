#!/usr/bin/env python3
"""tools/run_ring_rank.py rank world port nx rows steps [transport] — ONE rank of a row-partitioned run, started by hand
(tools/overlap_trace_ranks.sh starts `world` of them, each under its own rocprofv3; no launcher in between, so the profiled
program is the one after `--`).  The global grid is nx x (rows * world); rank r owns rows [r*rows, (r+1)*rows) on device
r mod (visible devices).  transport:
  rccl  halo rows by RCCL send/recv, velocity sums by RCCL all-reduce (one rank per device: the multi-GPU form)
  peer  halo rows by peer stores into the neighbours' HIP-IPC-mapped grids, no communicator (works with several ranks on
        ONE device too: the process boundaries are real, the xGMI hop is not)
The rendezvous (RCCL id / peer descriptors) goes over torch.distributed's gloo backend on 127.0.0.1:port."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, nx, rows, steps = (int(v) for v in sys.argv[1:7])
    transport = sys.argv[7] if len(sys.argv) > 7 else "peer"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import lbm_amd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = rank % max(1, torch.cuda.device_count())
    if world == 1:
        lbm_amd.set_default("force_halo", 1)
    ny = rows * world
    ob = np.zeros((ny, nx), np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    p = lbm_amd.make_params(nx, ny, steps + 16 + 64 + 8, obstacles=ob)   # warm-up + timed + profiled launch sets
    if transport == "rccl":
        box = [lbm_amd.comm_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        sim = lbm_amd.LBM(p, ob, rank=rank, nranks=world, device=dev, comm=box[0])
    else:
        sim = lbm_amd.LBM(p, ob, rank=rank, nranks=world, device=dev, comm=None)
        infos = [None] * world
        dist.all_gather_object(infos, sim.peer_info())
        sim.connect_peers(infos[(rank - 1) % world], infos[(rank + 1) % world])
    sim.upload(None)
    dist.barrier()
    sim.run(16)
    sim.sync()
    dist.barrier()
    ms = sim.run_timed(steps)
    sim.sync()
    t = torch.tensor([ms], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    st = sim.run_profiled(8 * max(1, sim.get_option("launch_steps")))
    print("rank %d/%d  %dx%d slab of %dx%d (%s, device %d, halo depth %d, %d steps per set): %.2f us/step (max over ranks), "
          "%.0f MLUPS whole job; launch set: edge %.1f us, exchange %.1f us, interior %.1f us, period %.1f us" % (
              rank, world, nx, rows, nx, ny, transport, dev, sim.get_option("halo_depth"), sim.get_option("launch_steps"),
              float(t[0]) / steps * 1e3, nx * ny * steps / float(t[0]) / 1e3, st["edge_us"], st["exchange_us"], st["interior_us"],
              st["set_period_us"]), flush=True)
    dist.barrier()
    if transport != "rccl":
        sim.disconnect_peers()        # unmap the neighbours' grids, meet, and only then free one's own
        dist.barrier()
    sim.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

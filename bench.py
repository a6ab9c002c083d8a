#!/usr/bin/env python3
"""bench.py — MLUPS and fraction of the HBM roofline of the D2Q9-BGK timestep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one lattice-Boltzmann timestep (accelerate_flow + stream + collide + av_vels
reduction, the reference's loop body d2q9-bgk.c:221-238) over the whole grid.  Workload: the
synthetic 8192x8192 lid-driven cavity of BASELINE.json (only the four border lines blocked),
uniform rest initial state, fp32.  With N > 1 the grid is row-partitioned over N ranks (one
process per GPU), halos go by RCCL send/recv inside liblbm_hip.so, the velocity sums by an RCCL
all-reduce; the RCCL id is distributed with torch.distributed.  Prints ONE JSON line on rank 0.

Algorithmic traffic: 72 B per lattice update (9 fp32 loads + 9 fp32 stores, kernels.cl:104-112,
189-197); one launch of the step kernel updates every cell of the rank's slab once.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_LU = 72.0
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cavity(nx, ny):
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    return ob


def make_workload(name, nx, ny):
    if name == "cavity":
        return cavity(nx, ny)
    if name == "empty":
        return np.zeros((ny, nx), dtype=np.int32)
    if name == "tiled":
        import lbm_amd
        _, ob = lbm_amd.read_inputs(os.path.join(ROOT, "inputs", "input_1024x1024.params"),
                                    os.path.join(ROOT, "inputs", "obstacles_1024x1024.dat"))
        assert nx % 1024 == 0 and ny % 1024 == 0
        return np.tile(ob, (ny // 1024, nx // 1024))
    raise ValueError(name)


def cpu_baseline(nx, ny, obstacles, accel, budget_s=15.0):
    """Serial fp32 oracle (the CPU restatement of the reference's timestep) on the host cores of
    this box: a bounded number of timesteps of the same grid, 1 thread."""
    from oracle.oracle import Oracle
    orc = Oracle("f32")
    p = orc.make_params(nx, ny, 1, 10, 0.1, accel, 1.85)
    orc.set_obstacles(p, obstacles)
    src = orc.init_cells(p)
    dst = np.empty_like(src)
    # one untimed step for page faults, then as many as fit the budget (at least 2)
    orc.accelerate_flow(p, src, obstacles)
    orc.timestep(p, src, dst, obstacles)
    src, dst = dst, src
    n, t0 = 0, time.perf_counter()
    while True:
        orc.accelerate_flow(p, src, obstacles)
        orc.timestep(p, src, dst, obstacles)
        src, dst = dst, src
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and (el > budget_s or n >= 64):
            break
    model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # the same oracle with OpenMP over this process's CPU share, as a second reference point (a few steps only)
    omp = None
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
        os.environ["OMP_NUM_THREADS"] = str(threads)
        orc_omp = Oracle("f32", omp=True)
        p_omp = orc_omp.make_params(nx, ny, 1, 10, 0.1, accel, 1.85)
        orc_omp.set_obstacles(p_omp, obstacles)
        orc_omp.timestep(p_omp, src, dst, obstacles)
        m, t1 = 0, time.perf_counter()
        while m < 2 or (time.perf_counter() - t1 < 4.0 and m < 64):
            orc_omp.accelerate_flow(p_omp, src, obstacles)
            orc_omp.timestep(p_omp, src, dst, obstacles)
            src, dst = dst, src
            m += 1
        omp = {"value": round(nx * ny * m / (time.perf_counter() - t1) / 1e6, 1), "cores": threads, "steps": m}
    except Exception:  # the baseline is informational; never fail the bench over it
        omp = None
    return {"value": round(nx * ny * n / el / 1e6, 2), "unit": "MLUPS", "cores": 1, "kind": "port", "openmp": omp,
            "sample": "%d timesteps of the same %dx%d grid with the serial fp32 oracle "
                      "(oracle/d2q9_oracle.c, gcc -O3 -march=native, 1 of %d host cores, %s)" % (n, nx, ny, os.cpu_count(), model)}


# ---- torch.distributed plumbing (also exercised with the gloo backend on CPU: tests/test_multirank_gloo.py) ----

def init_dist(backend, rank, world, device=None):
    """One process per GPU: MASTER_ADDR/MASTER_PORT come from the launcher (torch.distributed.run)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def share_comm_id(dist, rank, blob, nbytes, device):
    """Rank 0 made the RCCL unique id (lbm_comm_get_id); every rank gets the same bytes."""
    import torch
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if rank == 0:
        assert len(blob) == nbytes
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


def max_over_ranks(dist, values, device):
    import torch
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--nx", type=int, default=8192)
    ap.add_argument("--ny", type=int, default=8192)
    ap.add_argument("--workload", default="cavity", choices=["cavity", "empty", "tiled"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the nx x ny grid is split over the ranks; weak: every rank gets ny rows")
    ap.add_argument("--accel", type=float, default=0.005)
    ap.add_argument("--fuse", type=int, default=-1, help="timesteps per launch of the register/LDS-window kernels: 0, 1 (= 2), 3, 4; -1: library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 1024x1024 side measurement")
    args = ap.parse_args()

    import torch  # device plumbing + torch.distributed (RCCL) only
    import lbm_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed launch with that many ranks" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one GPU per rank; if the launcher narrowed the visible devices per rank, index within what is visible
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # LBM_BENCH_RANK_MODE=1 drives the one-process-per-GPU code path with a single rank: the rank is its own ring
    # neighbour (default "force_halo"), so torch.distributed + the library's RCCL communicator run on a one-GPU box
    rank_mode = world > 1 or os.environ.get("LBM_BENCH_RANK_MODE") == "1"
    if rank_mode and world == 1:
        lbm_amd.set_default("force_halo", 1)
    dist = init_dist("nccl", rank, world, torch.device("cuda", local_rank)) if rank_mode else None

    nx = args.nx
    ny = args.ny * (world if args.scaling == "weak" else 1)
    total_steps = args.warmup + args.steps
    obstacles = make_workload(args.workload, nx, ny)
    params = lbm_amd.make_params(nx, ny, total_steps, 10, 0.1, args.accel, 1.85, obstacles)

    if rank_mode:
        cid = share_comm_id(dist, rank, lbm_amd.comm_id() if rank == 0 else None,
                            lbm_amd.load_library().lbm_comm_id_size(), torch.device("cuda", local_rank))
        sim = lbm_amd.LBM(params, obstacles, rank=rank, nranks=world, device=local_rank, comm=cid)
    else:
        sim = lbm_amd.LBM(params, obstacles)
    if args.fuse >= 0:
        sim.set_option("fuse", args.fuse)
    fused = {0: 0, 1: 2, 3: 3, 4: 4}[sim.get_option("fuse")]   # timesteps per launch of the dominant kernel (0: one)
    sim.upload(None)  # uniform rest state, built on the device
    y0, y1 = sim.row_range()

    def fence():
        sim.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    sim.run(args.warmup)
    fence()
    t0 = time.perf_counter()
    loop_ms = sim.run_timed(args.steps)   # HIP events on the stream the step kernels run on
    fence()
    wall = time.perf_counter() - t0
    if dist is not None:
        wall, loop_ms = max_over_ranks(dist, [wall, loop_ms], torch.device("cuda", local_rank))

    # sanity on the result of the timed run: finite, positive average velocity on every rank
    _, av = sim.download(cells=False)
    ok = bool(np.all(np.isfinite(av)) and av[-1] > 0)

    out = None
    if rank == 0:
        lups = nx * ny * args.steps / wall
        rows_local = y1 - y0
        # the dominant kernel advances `steps_per_launch` timesteps of the rank's slab per launch
        steps_per_launch = fused if fused else 1
        launches = args.steps // steps_per_launch + args.steps % steps_per_launch
        launch_s = loop_ms * 1e-3 / launches
        alg_bytes = BYTES_PER_LU * nx * rows_local * steps_per_launch
        achieved = alg_bytes / launch_s / 1e9
        out = {
            "metric": "MLUPS", "value": round(lups / 1e6, 1), "unit": "MLUPS (million lattice updates/s)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 5), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d %s, D2Q9-BGK fused timestep, uniform rest start" % (nx, ny, {
                "cavity": "lid-driven cavity (4 border lines blocked)", "empty": "no obstacles (periodic)",
                "tiled": "obstacles_1024x1024 tiled up"}[args.workload]),
                "nx": nx, "ny": ny, "rows_per_gpu": rows_local, "partition": "rows x%d" % world,
                "omega": 1.85, "accel": args.accel, "density": 0.1},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": {0: "d2q9_step", 2: "d2q9_step2 (two timesteps per launch)",
                                    3: "d2q9_step3 (three timesteps per launch)",
                                    4: "d2q9_step4 (four timesteps per launch)"}[fused],
                         "launch_us": round(launch_s * 1e6, 2), "steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "result_ok": ok,
        }
        # measured PMC traffic of the same command, when a profile of this workload is committed
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1:
            with open(tp) as f:
                tj = json.load(f)
            key = "%dx%d/step%d" % (nx, ny, steps_per_launch)
            if key in tj:
                out["roofline"]["traffic"] = tj[key]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_frac"] = round(tj[key]["hbm_bytes_per_launch"] / launch_s / 1e9 / HBM_PEAK_GBPS, 4)
                out["roofline"]["traffic_source"] = tj[key].get("source")
        # what a plain float4 copy achieves on this box right now (context for `frac`; the spec peak stays `peak`)
        try:
            out["roofline"]["copy_kernel_gbps"] = round(lbm_amd.copy_bandwidth_gbps(1 << 30, 10), 1)
        except lbm_amd.LBMError:
            pass
        if fused:
            out["roofline"]["note"] = ("frac uses the ALGORITHMIC 72 B per lattice update; the multi-step kernels keep the "
                                       "intermediate states in registers and really move far fewer bytes per update (traffic), "
                                       "so frac can exceed 1 while traffic_frac is the share of the 8 TB/s peak actually used")
    sim.close()

    if world == 1 and rank == 0:
        if not args.no_extra and (nx, ny) != (1024, 1024):
            # the reference's own largest input, for the 1024x1024 figure the north star asks for
            p2, ob2 = lbm_amd.read_inputs(os.path.join(ROOT, "inputs", "input_1024x1024.params"),
                                          os.path.join(ROOT, "inputs", "obstacles_1024x1024.dat"))
            n2 = 4000
            p2.max_iters = n2 + 200
            with lbm_amd.LBM(p2, ob2) as s2:
                s2.upload(None)
                s2.run(200)
                s2.sync()
                t1 = time.perf_counter()
                ms2 = s2.run_timed(n2)
                s2.sync()
                w2 = time.perf_counter() - t1
            out["also"] = {"workload": "input_1024x1024.params + obstacles_1024x1024.dat (fits the 256 MiB Infinity Cache)",
                           "value": round(1024 * 1024 * n2 / w2 / 1e6, 1), "unit": "MLUPS", "steps": n2,
                           "roofline_frac_if_hbm": round(BYTES_PER_LU * 1024 * 1024 / (ms2 * 1e-3 / n2) / 1e9 / HBM_PEAK_GBPS, 4)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, obstacles, args.accel)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

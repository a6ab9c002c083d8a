#!/usr/bin/env python3
"""bench.py — MLUPS and fraction of the HBM roofline of the D2Q9-BGK timestep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own ranks: before anything touches the GPU
it runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (one rank per GPU over
RCCL), relays rank 0's JSON line and exits with the child's code (--launcher one-process: the lbm_create(ndev=N) form
of INTEGRATION.md section 3 instead, one process driving N slabs; the line says which).

A "step" is one lattice-Boltzmann timestep (accelerate_flow + stream + collide + av_vels reduction, the reference's
loop body d2q9-bgk.c:221-238) over the whole grid.  Workload: the synthetic 8192x8192 lid-driven cavity of
BASELINE.json (only the four border lines blocked), uniform rest initial state, fp32.  With N > 1 the grid is
row-partitioned over N ranks (one process per GPU); the halo rows move inside liblbm_hip.so — by peer stores over
xGMI into the neighbours' HIP-IPC-mapped grids and by RCCL send/recv, both measured back to back, the faster one is
`value` — and the velocity sums by an RCCL all-reduce; the RCCL id and the peer descriptors are distributed with
torch.distributed.  Prints ONE JSON line on rank 0 — with N ranks two: the headline record (`"partial"` set) as soon as the
timed leg and its check are done, and the full record, which repeats it and adds the side legs, as the LAST line: a run cut
off at somebody's time limit still leaves a valid record behind.

Before anything is timed on N ranks, every halo transport has to pass three checks (`transport_check`): the reference's
1024x1024 obstacles over the ranks for 400 steps against the oracle (>= 50 exchanges, the small-slab kernels), an
8192 x 768N grid for 40 steps against the oracle (the deep window kernels with their in-kernel push / two-stream RCCL
launch sets: what the headline leg runs), and on the timed geometry itself 32 steps from a seeded state whose per-slab
digests must be bit-identical between the transports (or to the undivided grid).  A transport that fails one of them is
neither timed nor reported.

What the roofline object means (every field is recomputable from the others and from profiles/):
  a launch of the dominant kernel advances S timesteps and must read the grid once and write it once, whatever S is:
  model bytes per launch = (72 + 1) B x cells (9 fp32 in, 9 fp32 out, one mask byte)         -> achieved, frac (<= 1)
  algorithmic bytes      = 72 B per lattice update x S x cells (kernels.cl:104-112,189-197)
                           -> roofline.algorithmic, can exceed the HBM peak: temporal blocking moves 1/S of them
  traffic                = HBM bytes per launch counted by rocprofv3 (profiles/traffic.json: another run, named there)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_LU = 72.0       # algorithmic: 9 fp32 loads + 9 fp32 stores per lattice update
MASK_BYTES = 1.0          # obstacle mask, one byte per cell and launch
HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
KERNELS = {0: "d2q9_step", 2: "d2q9_step2 (two timesteps per launch)", 3: "d2q9_step3 (three timesteps per launch)",
           4: "d2q9_step4 (four timesteps per launch)",
           6: "d2q9_deep (up to 6 timesteps per launch, lanes of two cells)",
           7: "d2q9_deep (up to 7 timesteps per launch, lanes of two cells)",
           8: "d2q9_deep (up to 8 timesteps per launch, lanes of two cells)"}


def cavity(nx, ny):
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    return ob


def shipped(size):
    import lbm_amd
    return lbm_amd.read_inputs(os.path.join(ROOT, "inputs", "input_%s.params" % size),
                               os.path.join(ROOT, "inputs", "obstacles_%s.dat" % size))


def make_workload(name, nx, ny):
    if name == "cavity":
        return cavity(nx, ny)
    if name == "empty":
        return np.zeros((ny, nx), dtype=np.int32)
    if name == "tiled":
        _, ob = shipped("1024x1024")
        assert nx % 1024 == 0 and ny % 1024 == 0
        return np.tile(ob, (ny // 1024, nx // 1024))
    raise ValueError(name)


T_START = time.time()


def log(msg, rank=0):
    """progress on stderr (rank 0): a long multi-rank run says where it is, with the seconds since it started"""
    if rank == 0:
        print("bench.py [%6.1f s] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


def seeded_state(nx, rows_global, density=0.1, amp=0.2, period=61, seed=2026):
    """Deterministic non-equilibrium state of the given GLOBAL rows, float32[9, len(rows), nx]: w_k * density * (1 +- amp/2),
    an integer hash of (speed, row mod 61, column).  Any rank computes exactly the rows it needs and any two ranks the same
    values for the same row; 61 divides no slab height, so the rows either side of every slab boundary differ from each other
    and from what they were a few steps earlier — a stale or misplaced halo row cannot pass for the right one."""
    rows_global = np.asarray(rows_global, dtype=np.int64)
    mask = np.uint64(0xFFFFFFFF)
    x = np.arange(nx, dtype=np.uint64)[None, :]
    r = np.arange(period, dtype=np.uint64)[:, None]
    w = [4.0 / 9.0] + [1.0 / 9.0] * 4 + [1.0 / 36.0] * 4
    tile = np.empty((9, period, nx), dtype=np.float32)
    for k in range(9):
        h = (r * np.uint64(0x9E3779B1) + x * np.uint64(0x85EBCA77) + np.uint64(((k + 1) * 0xC2B2AE3D + seed) & 0xFFFFFFFF)) & mask
        h ^= h >> np.uint64(15)
        h = (h * np.uint64(0x2C1B3C6D)) & mask
        h ^= h >> np.uint64(12)
        h = (h * np.uint64(0x297A2D39)) & mask
        h ^= h >> np.uint64(15)
        tile[k] = (w[k] * density * (1.0 + amp * (h.astype(np.float64) / 4294967296.0 - 0.5))).astype(np.float32)
    return np.ascontiguousarray(tile[:, rows_global % period, :])


def band_oracle(orc, nx, ny, obstacles, y0, y1, nsteps, density, accel, omega, state_fn=seeded_state):
    """The fp32 oracle for global rows [y0, y1) of an nx x ny periodic grid over nsteps timesteps, computed on a BAND of rows:
    the rows themselves plus nsteps + 1 rows either side — their domain of dependence (kernels.cl:104-112: one row per
    step); what the band's own periodic wrap lets in at its two ends travels one row per step and stops short of them.
    Returns (cells float32[9, y1 - y0, nx] after nsteps, raw float64[nsteps]: the sum of |j|/rho over these rows' fluid
    cells per step, kernels.cl:198 before the division by the free cells).  A rank of an N-rank check pays for its own
    slab, not for the whole grid."""
    rows, margin = y1 - y0, nsteps + 1
    if rows + 2 * margin >= ny:
        band, off = np.arange(ny), y0
    else:
        band, off = np.arange(y0 - margin, y1 + margin) % ny, margin
    nb = len(band)
    src = np.ascontiguousarray(state_fn(nx, band), dtype=np.float32)
    dst = np.empty_like(src)
    obb = np.ascontiguousarray(obstacles[band], dtype=np.int32)
    po = orc.make_params(nx, nb, max(nsteps, 1), 10, density, accel, omega)
    accel_rows = [int(i) for i in np.nonzero(band == ny - 2)[0]]   # kernels.cl:18: global row ny-2, if the band holds it
    raw = np.zeros(nsteps, dtype=np.float64)
    for t in range(nsteps):
        for a in accel_rows:
            orc.accelerate_row(po, src, obb, a)
        if off > 0:
            orc.timestep_rows(po, src, dst, obb, 0, off)
        raw[t] = orc.timestep_rows(po, src, dst, obb, off, off + rows)
        if off + rows < nb:
            orc.timestep_rows(po, src, dst, obb, off + rows, nb)
        src, dst = dst, src
    return src[:, off:off + rows], raw


# ---- CPU baseline: the oracle (CPU restatement of the reference's timestep) on this box's host cores ----------

def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def oracle_rate(precision, nx, ny, obstacles, accel, budget_s, max_steps, omp=False, av_out=None):
    """(steps, seconds) of one oracle build on an nx x ny grid from the uniform rest state: at least 2 steps, at most
    max_steps / budget_s.  av_out (a list) receives av_vels[0..] of the steps taken, the untimed first one included —
    the record bench.py holds the timed GPU context's av_vels against (`result_check`)."""
    from oracle.oracle import Oracle
    orc = Oracle(precision, omp=omp)
    p = orc.make_params(nx, ny, 1, 10, 0.1, accel, 1.85)
    orc.set_obstacles(p, obstacles)
    src = orc.init_cells(p)
    dst = np.empty_like(src)
    orc.accelerate_flow(p, src, obstacles)   # one untimed step for page faults
    av = [orc.timestep(p, src, dst, obstacles)]
    src, dst = dst, src
    n, t0 = 0, time.perf_counter()
    while True:
        orc.accelerate_flow(p, src, obstacles)
        av.append(orc.timestep(p, src, dst, obstacles))
        src, dst = dst, src
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and (el > budget_s or n >= max_steps):
            break
    if av_out is not None:
        av_out.extend(av)
    return n, el


def cpu_baseline(nx, ny, obstacles, accel, av_out=None):
    """Serial fp32 oracle, 1 thread, on a bounded sample of the headline workload (`value`), plus the other figures
    SURVEY.md 8(d) asks for: the 128x128 input run to the end and sent through the checker (BASELINE config 1), the
    rate on the 1024x1024 input (200 steps) in fp32 and fp64, and an OpenMP figure over this process's CPU share.
    About 35 s of CPU work in all."""
    model, ncpu = cpu_model(), os.cpu_count()
    n, el = oracle_rate("f32", nx, ny, obstacles, accel, 9.0, 64, av_out=av_out)
    out = {"value": round(nx * ny * n / el / 1e6, 2), "unit": "MLUPS", "cores": 1, "kind": "port",
           "sample": "%d timesteps of the same %dx%d grid with the serial fp32 oracle (oracle/d2q9_oracle.c, gcc -O3 "
                     "-march=native, 1 of %d host cores, %s)" % (n, nx, ny, ncpu, model)}
    # BASELINE config 1: input_128x128 on the serial CPU path, full length, check.py must pass
    try:
        import gzip
        import io
        import shutil
        from check.check import run_check
        with tempfile.TemporaryDirectory() as d:
            files = [os.path.join(ROOT, "inputs", f) for f in ("input_128x128.params", "obstacles_128x128.dat")]
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(ROOT, "oracle", "d2q9-bgk-serial-f32")] + files, cwd=d, capture_output=True,
                               text=True, timeout=120)
            wall = time.perf_counter() - t0
            el128 = float([ln for ln in r.stdout.splitlines() if ln.startswith("Elapsed time:")][0].split()[2])
            refs = []
            for name in ("128x128.av_vels.dat", "128x128.final_state.dat"):
                dst = os.path.join(d, "ref_" + name)
                with gzip.open(os.path.join(ROOT, "tests", "golden", "check", name + ".gz"), "rb") as fi, open(dst, "wb") as fo:
                    shutil.copyfileobj(fi, fo)
                refs.append(dst)
            code, avd, fsd = run_check(refs[0], refs[1], os.path.join(d, "av_vels.dat"), os.path.join(d, "final_state.dat"),
                                       1.0, io.StringIO())
        out["input_128x128_full_run"] = {
            "value": round(128 * 128 * 40000 / el128 / 1e6, 2), "unit": "MLUPS", "steps": 40000, "elapsed_s": round(el128, 3),
            "wall_s_with_file_output": round(wall, 2), "precision": "f32", "check_py": "passed" if code == 0 else "FAILED",
            "max_diff_pcnt": {"av_vels": round(abs(avd["max_diff_pcnt"]), 4), "final_state": round(abs(fsd["max_diff_pcnt"]), 4)}}
    except Exception as e:  # the baseline is informational; never fail the bench over it
        out["input_128x128_full_run"] = {"error": str(e)[:200]}
    try:
        _, ob1 = shipped("1024x1024")
        n32, e32 = oracle_rate("f32", 1024, 1024, ob1, 0.01, 60.0, 200)
        n64, e64 = oracle_rate("f64", 1024, 1024, ob1, 0.01, 60.0, 200)
        out["input_1024x1024_rate"] = {"f32": {"value": round(1024 * 1024 * n32 / e32 / 1e6, 2), "steps": n32},
                                       "f64": {"value": round(1024 * 1024 * n64 / e64 / 1e6, 2), "steps": n64}, "unit": "MLUPS", "cores": 1}
    except Exception as e:
        out["input_1024x1024_rate"] = {"error": str(e)[:200]}
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
        os.environ["OMP_NUM_THREADS"] = str(threads)
        m, em = oracle_rate("f32", nx, ny, obstacles, accel, 4.0, 64, omp=True)
        out["openmp"] = {"value": round(nx * ny * m / em / 1e6, 1), "cores": threads, "steps": m, "unit": "MLUPS"}
    except Exception:
        out["openmp"] = None
    return out


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


def gather_blobs(dist, blob, world, device):
    """all_gather of one equally sized byte blob per rank (peer descriptors; per-rank statistics as padded JSON)"""
    import torch
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [bytes(p.cpu().numpy().tobytes()) for p in parts]


def max_over_ranks(dist, values, device):
    import torch
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def sum_over_ranks(dist, values, device):
    import torch
    t = torch.tensor(np.asarray(values, dtype=np.float64), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def min_over_ranks(dist, value, device):
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0])


class TransportFailed(RuntimeError):
    """raised on EVERY rank alike (the ranks vote) when a halo transport failed on any of them"""


class Env:
    """what every leg needs: the library binding, torch.distributed and this rank's place in the job"""

    def __init__(self, lbm_amd, torch, dist, rank, world, local_rank, device, rehearsal=False):
        self.lbm, self.torch, self.dist = lbm_amd, torch, dist
        self.rank, self.world, self.local_rank, self.device = rank, world, local_rank, device
        # rehearsal (LBM_BENCH_REHEARSAL=1): N ranks as N processes on fewer GPUs than ranks — torch.distributed over gloo,
        # contexts without RCCL communicator (RCCL refuses two ranks per device), halo rows by peer stores over HIP IPC,
        # velocity records added up here.  Exercises every N > 1 code path of this file on a one-GPU box; its numbers
        # are NOT a benchmark (the ranks share a GPU) and the line says so.
        self.rehearsal = rehearsal

    def all_ok(self, ok):
        """the ranks' vote: True only if every rank says so (an all-reduce, so it also lines the ranks up)"""
        return min_over_ranks(self.dist, 1.0 if ok else 0.0, self.device) > 0.5


PEER_TIMEOUT_MS = 5000   # bench runs: a rank waits this long for a neighbour's halo rows (library default: 30 s)


class RankSim:
    """One rank's slab of a row-partitioned grid on ONE halo transport ("rccl": grouped send/recv; "peer": stores into
    the neighbours' HIP-IPC-mapped grids).  Every rank takes part in every collective in here whatever happened to
    it — a failed descriptor travels as a zero blob, a failed mapping as a 'no' vote — so that no rank is left
    blocked in a collective the others skipped."""

    def __init__(self, env, params, obstacles, transport):
        lbm_amd, dist = env.lbm, env.dist
        cid = None
        if not env.rehearsal:
            cid = share_comm_id(dist, env.rank, lbm_amd.comm_id() if env.rank == 0 else None,
                                lbm_amd.load_library().lbm_comm_id_size(), env.device)
        self.sim = lbm_amd.LBM(params, obstacles, rank=env.rank, nranks=env.world, device=env.local_rank, comm=cid)
        self.env = env
        self.transport = transport
        self.error = None
        if transport == "rccl2":
            # RCCL with the launch sets of round 3: edge launch + exchange + interior launch on two streams (option "compact" 0) — the
            # form main() falls back to when the staged launch sets (one launch per set, exchange behind a stream wait-value) fail a check
            self.sim.set_option("compact", 0)
        if transport == "peer":
            size = lbm_amd.load_library().lbm_peer_info_size()
            try:
                blob = self.sim.peer_info()
            except lbm_amd.LBMError as e:
                blob, self.error = bytes(size), str(e)
            infos = gather_blobs(dist, blob, env.world, env.device)
            if all(any(b) for b in infos):          # the same answer on every rank
                try:
                    self.sim.connect_peers(infos[(env.rank - 1) % env.world], infos[(env.rank + 1) % env.world])
                    if cid is not None:
                        self.sim.set_option("transport", 3)   # (a context with a communicator stays on RCCL until told)
                    self.sim.set_option("halo_timeout_ms", PEER_TIMEOUT_MS)
                except lbm_amd.LBMError as e:
                    self.error = str(e)
            elif not self.error:
                self.error = "a rank of the ring has no peer descriptor"
            if not env.all_ok(self.error is None):
                self.error = self.error or "another rank could not map its ring neighbours"
                self.close()
                raise TransportFailed("peer: " + self.error)

    def av_vels(self):
        """the global av_vels record on every rank (a collective: every rank calls it)"""
        av = self.sim.download(cells=False)[1]
        if self.env.rehearsal:   # no communicator: the library returned this rank's own sums, the caller adds the ranks'
            t = self.env.torch.from_numpy(av.astype(np.float64))
            self.env.dist.all_reduce(t)
            av = t.numpy().astype(np.float32)
        return av

    def close(self):
        """Collective: every rank unmaps its neighbours' grids, the ranks meet, and only then does anybody free its own —
        device memory that another process still has mapped through HIP IPC must not be freed (undefined behaviour; a first
        version that closed rank by rank failed once in three full test-suite runs on some boxes)."""
        if self.sim is not None:
            if self.transport == "peer":
                try:
                    self.sim.disconnect_peers()
                except self.env.lbm.LBMError:
                    pass
                self.env.all_ok(True)
            self.sim.close()
            self.sim = None

    def abandon(self):
        """non-collective: this rank is leaving through an unexpected exception (the launcher will end the job); no vote —
        the other ranks are not at the meeting point"""
        if self.sim is not None:
            try:
                self.sim.close()
            finally:
                self.sim = None


def synced(env, sim):
    """lbm_sync on every rank, then the vote: a transport that failed on one rank (LBM_ERR_COMM: a neighbour's halo
    rows never came, an RCCL error) is a failure on all of them — nobody walks into the next collective alone"""
    err = None
    try:
        sim.sync()
        env.torch.cuda.synchronize()
    except env.lbm.LBMError as e:
        err = str(e)
    if env.dist is None:
        if err:
            raise TransportFailed(err)
        return
    if not env.all_ok(err is None):
        raise TransportFailed(err or "another rank's halo transport failed")


def timed_run(env, sim, warmup, steps):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
    sim.run(warmup)
    synced(env, sim)           # sync + all-reduce vote = the barrier of the contract
    t0 = time.perf_counter()
    loop_ms = sim.run_timed(steps)   # HIP events on the stream the step kernels run on
    synced(env, sim)
    wall = time.perf_counter() - t0
    if env.dist is not None:
        wall, loop_ms = max_over_ranks(env.dist, [wall, loop_ms], env.device)
    return wall, loop_ms


def profile_all_ranks(env, sim, nsteps, extra):
    """lbm_run_profiled on every rank, gathered on all: where a launch set's time goes (edge / exchange / interior)"""
    st = sim.run_profiled(nsteps)
    st = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}
    st.update(rank=env.rank, **extra)
    blob = json.dumps(st).encode().ljust(1024)[:1024]
    return [json.loads(b.decode().strip()) for b in gather_blobs(env.dist, blob, env.world, env.device)]


def rank_leg(env, params, obstacles, transports, warmup, steps, fuse=-1, profile=True, want_av=False, before=None):
    """One workload over the ranks, once per halo transport — a fresh context each, so that a transport that fails
    leaves nothing behind for the next.  Returns {transport: {...}}; a failed transport has an "error" instead of times."""
    runs = {}
    for tr in transports:
        entry, rs = {}, None
        try:
            rs = RankSim(env, params, obstacles, tr)
            sim = rs.sim
            if fuse >= 0:
                sim.set_option("fuse", fuse)
            sim.upload(None)  # uniform rest state, built on the device
            if before:
                before()      # the declared pre-warm-up (calibration launches), right before the W warm-up steps
            entry["wall_s"], entry["loop_ms"] = timed_run(env, sim, warmup, steps)
            entry["row_range"] = sim.row_range()
            entry["options"] = {k: sim.get_option(k) for k in OPTION_KEYS}
            if want_av:
                entry["av"] = rs.av_vels()     # collective (RCCL all-reduce): every rank got here
            if profile:
                per = max(entry["options"]["multistep"], entry["options"]["launch_steps"], 1)
                entry["per_rank"] = profile_all_ranks(env, sim, 8 * per, dict(rows=entry["row_range"][1] - entry["row_range"][0]))
                synced(env, sim)
        except TransportFailed as e:   # raised on every rank alike (the ranks voted): the collective close below is safe
            entry = {"error": str(e)[:300]}
            print("bench.py: rank %d: transport %s failed on a %dx%d grid: %s" % (env.rank, tr, params.nx, params.ny, str(e)[:300]), file=sys.stderr, flush=True)
        except BaseException:
            if rs is not None:
                rs.abandon()
            raise
        if rs is not None:
            rs.close()
        runs[tr] = entry
    return runs


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def fault_legs(transport=None):
    """TEST HOOK: LBM_BENCH_FAULT=stale_halo[@leg,leg][:transport] makes one halo exchange of the named check legs ("small",
    "deep", "digest"; none named = all three) deliver nothing (library option debug_stale_exchange) — the checks must catch it;
    with `:transport` only the contexts of that transport are hit (how the fall-back from staged to two-stream RCCL sets is tested)"""
    f = os.environ.get("LBM_BENCH_FAULT", "")
    if not f.startswith("stale_halo"):
        return set()
    only = None
    if ":" in f:
        f, only = f.rsplit(":", 1)
    if transport is not None and only is not None and transport != only:
        return set()
    return set(f.split("@", 1)[1].split(",")) if "@" in f else {"small", "deep", "digest"}


def upload_rows(sim, nx, ny, y0, y1, rows):
    """this rank's rows of a global initial state: the ABI takes the GLOBAL float[9][ny][nx] array and reads rows [y0, y1) of
    it — the other rows are never touched, so they are never committed (np.empty: untouched pages cost nothing)"""
    cells = np.empty((9, ny, nx), dtype=np.float32)
    cells[:, y0:y1] = rows
    sim.upload(cells)


def oracle_leg(env, label, p, ob, transports, nsteps, fault):
    """One check leg against the ORACLE: the nx x ny grid of `p` row-partitioned over the ranks, a seeded non-equilibrium state
    (every slab boundary carries non-trivial data from the first step on, unlike a cavity at rest), nsteps timesteps on every
    transport.  Each rank holds its own rows of every distribution (<= 2e-5 relative) and the all-reduced av_vels record
    (<= 1e-4) against the fp32 oracle (oracle/d2q9_oracle.c restating kernels.cl:9-231) of ITS rows, computed on its own host
    cores on a band of rows (band_oracle)."""
    from oracle.oracle import Oracle
    nx, ny = p.nx, p.ny
    y0, rows = env.lbm.slab_rows(ny, env.world, env.rank)
    y1 = y0 + rows
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(8, len(os.sched_getaffinity(0)) // max(1, env.world))))
    orc = Oracle("f32", omp=True)
    t0 = time.perf_counter()
    ref, raw = band_oracle(orc, nx, ny, ob, y0, y1, nsteps, p.density, p.accel, p.omega)
    raw_all = sum_over_ranks(env.dist, raw, env.device) if env.dist is not None else raw
    av_ref = (raw_all * float(p.free_cells_inv)).astype(np.float32)     # kernels.cl:202: sum * FREE_CELLS_INV
    t_oracle = time.perf_counter() - t0
    state0 = seeded_state(nx, np.arange(y0, y1))
    out = {}
    for tr in transports:
        res, rs = {}, None
        try:
            rs = RankSim(env, p, ob, tr)
            assert rs.sim.row_range() == (y0, y1)
            if fault and label in fault_legs(tr):
                rs.sim.set_option("debug_stale_exchange", 3)
            upload_rows(rs.sim, nx, ny, y0, y1, state0)
            rs.sim.run(nsteps)
            synced(env, rs.sim)
            got = rs.sim.download(av_vels=False)[0]
            av = rs.av_vels()
            ec, ea = max_rel(got[:, y0:y1], ref), max_rel(av, av_ref)
            del got
            ec, ea = max_over_ranks(env.dist, [ec, ea], env.device)
            o = {k: rs.sim.get_option(k) for k in OPTION_KEYS + ("compact",)}
            res = {"ok": bool(ec < 2e-5 and ea < 1e-4), "cells_max_rel": float("%.3g" % ec), "av_vels_max_rel": float("%.3g" % ea),
                   "halo_depth": o["halo_depth"], "exchanges": -(-nsteps // max(1, o["launch_steps"])),
                   "kernel": kernel_label(o).split(" (")[0] + (" x%d" % o["launch_steps"] if o["fuse"] >= 5 and not o["multistep"] else "") +
                             (", compact launch sets" if o["compact"] else ", two streams")}
            if tr == "peer":
                res["push_release"] = rs.sim.get_option("push_release")
        except TransportFailed as e:
            res = {"ok": False, "error": str(e)[:300]}
            print("bench.py: rank %d: transport %s failed the oracle check (%s): %s" % (env.rank, tr, label, str(e)[:300]), file=sys.stderr, flush=True)
        except BaseException:
            if rs is not None:
                rs.abandon()
            raise
        if rs is not None:
            rs.close()
        out[tr] = res
        log("transport_check %s: %s %s" % (label, tr, json.dumps(res)), env.rank)
    return {"workload": "%s: %dx%d, rows x%d, seeded non-equilibrium state, %d timesteps vs the fp32 oracle of each rank's rows (cells <= 2e-5, "
                        "av_vels <= 1e-4 relative)" % (label, nx, ny, env.world, nsteps),
            "oracle_s": round(t_oracle, 2), "transports": out}


def slab_digest(cells, y0, y1):
    """128-bit digest of rows [y0, y1) of all nine planes of a float32[9][ny][nx] array"""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for k in range(9):
        h.update(np.ascontiguousarray(cells[k, y0:y1]).data)
    return h.digest()


def digest_leg(env, params, obstacles, transports, nsteps, fault):
    """The check on the TIMED geometry itself, bit for bit and without an oracle: every transport advances the timed grid
    nsteps timesteps from the same seeded state, every rank digests its slab, the digests are gathered.  All kernels inline
    one collision and a cell's result does not depend on which slab or launch computed it, so the slabs' digests must be
    IDENTICAL between the transports — and identical to the digests of the same rows of the undivided grid, which rank 0
    computes on its own GPU whenever there are not two transports to hold against each other, or they disagree."""
    lbm_amd = env.lbm
    nx, ny = params.nx, params.ny
    y0, rows = lbm_amd.slab_rows(ny, env.world, env.rank)
    y1 = y0 + rows
    state0 = seeded_state(nx, np.arange(y0, y1))
    digests, out = {}, {}
    for tr in transports:
        rs, res = None, {}
        try:
            rs = RankSim(env, params, obstacles, tr)
            if fault and "digest" in fault_legs(tr):
                rs.sim.set_option("debug_stale_exchange", 3)
            upload_rows(rs.sim, nx, ny, y0, y1, state0)
            rs.sim.run(nsteps)
            synced(env, rs.sim)
            got = rs.sim.download(av_vels=False)[0]
            d = slab_digest(got, y0, y1)
            del got
            digests[tr] = gather_blobs(env.dist, d, env.world, env.device)
            res = {"ok": True}
        except TransportFailed as e:
            res = {"ok": False, "error": str(e)[:300]}
        except BaseException:
            if rs is not None:
                rs.abandon()
            raise
        if rs is not None:
            rs.close()
        out[tr] = res
    names = list(digests)
    agree = len(names) >= 2 and all(digests[t] == digests[names[0]] for t in names[1:])
    reference = None
    if names and not agree:
        # the undivided grid on rank 0's GPU: one slab, no halo rows, no transport (the other ranks wait in the gather)
        blob = bytes(16 * env.world)
        if env.rank == 0:
            with lbm_amd.LBM(params, obstacles) as sim:
                sim.upload(seeded_state(nx, np.arange(ny)))
                sim.run(nsteps)
                whole = sim.download(av_vels=False)[0]
            parts = []
            for r in range(env.world):
                r0, rn = lbm_amd.slab_rows(ny, env.world, r)
                parts.append(slab_digest(whole, r0, r0 + rn))
            blob = b"".join(parts)
            del whole
        blob = gather_blobs(env.dist, blob, env.world, env.device)[0]
        reference = [blob[16 * r:16 * r + 16] for r in range(env.world)]
    for tr in names:
        bad = [r for r in range(env.world) if digests[tr][r] != (reference[r] if reference is not None else digests[names[0]][r])]
        out[tr] = {"ok": not bad, "digest_rank0": digests[tr][0].hex()}
        if bad:
            out[tr]["ranks_that_differ"] = bad
            print("bench.py: rank %d: transport %s: the slabs of ranks %s differ from %s after %d steps on the timed geometry"
                  % (env.rank, tr, bad, "the undivided grid" if reference is not None else names[0], nsteps), file=sys.stderr, flush=True)
    log("transport_check digest: %s" % json.dumps(out), env.rank)
    return {"workload": "the timed %dx%d grid, rows x%d, seeded non-equilibrium state, %d timesteps: per-slab digests (blake2b-128 of "
                        "the nine planes) identical %s" % (nx, ny, env.world, nsteps,
                        "to the undivided grid's (one slab on rank 0)" if reference is not None else "between the transports"),
            "compared_with": "undivided grid" if reference is not None else "each other", "transports": out}


def transport_check(env, params, obstacles, transports):
    """Every halo transport through three checks before any of them is timed or chosen (module docstring): `small` — the
    reference's 1024x1024 obstacles for 400 steps (>= 50 exchanges of up to 8 halo rows; at 8 GPUs 128-row slabs on the
    LDS-tile kernel and its edge-tile push), `deep` — 8192 x 768N cells with side walls for 40 steps (slabs of 6M cells: the
    deep window kernels in the form the 8192x8192 leg runs them at 4 and 8 GPUs — one round of chunk pairs with push_chunk_pairs
    and the per-wave publisher over peer stores, the same launch with staging blocks and a stream wait-value over RCCL ("rccl2": the
    lone kernel on two streams) —, five exchanges of 8 rows; the record names the kernel that ran), both against the oracle; `digest` — the timed geometry, 32 steps,
    bit for bit.  A transport is `ok` only if it passed all three."""
    lbm_amd = env.lbm
    faults = fault_legs()
    legs = {}
    p2, ob2 = shipped("1024x1024")
    p2.max_iters = 400 + 8
    legs["small"] = oracle_leg(env, "small", p2, ob2, transports, 400, "small" in faults)
    nyd = 768 * env.world
    obd = np.zeros((nyd, 8192), dtype=np.int32)
    obd[:, 0] = obd[:, -1] = 1
    pd = lbm_amd.make_params(8192, nyd, 48, 10, 0.1, 0.005, 1.85, obd)
    legs["deep"] = oracle_leg(env, "deep", pd, obd, transports, 40, "deep" in faults)
    del obd
    legs["digest"] = digest_leg(env, params, obstacles, transports, 32, "digest" in faults)
    verdict = {}
    for tr in transports:
        ok = all(legs[k]["transports"].get(tr, {}).get("ok", False) for k in legs)
        verdict[tr] = dict(legs["small"]["transports"].get(tr, {}), ok=ok,
                           failed_legs=[k for k in legs if not legs[k]["transports"].get(tr, {}).get("ok", False)])
    return {"transports": verdict, "legs": legs, "fault_injected": sorted(faults) if faults else None}


def result_check(av_gpu, av_oracle, tol=1e-4):
    """`result_ok`: the timed context's av_vels[0..n) against the oracle's for the same steps from the same rest state
    on the same grid (kernels.cl:198 — every cell's |u| enters every step's entry: a launch that skipped levels, rows
    or strips shows here)"""
    n = min(len(av_gpu), len(av_oracle))
    if n < 2 or not np.all(np.isfinite(av_gpu)) or not av_gpu[-1] > 0:
        return False, {"compared_steps": n, "error": "no finite positive av_vels record to compare"}
    dev = max_rel(av_gpu[:n], av_oracle[:n])
    return bool(dev < tol), {"compared_steps": n, "av_vels_max_rel_vs_oracle": float("%.3g" % dev), "tolerance": tol,
                             "what": "av_vels[0..%d) of the timed context vs the fp32 oracle stepped from the same rest state" % n}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher around it: start the ranks as a child process.  This process never
    imports the HIP library and never touches the GPU (a process that has may not be replaced or forked safely on this
    pool).  It STREAMS the child's output: every record line ({"metric"...) goes to stdout the moment rank 0 prints it — the
    headline record first, the full record last — everything else to stderr; the exit code is the child's."""
    import threading
    env = dict(os.environ, LBM_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               PYTHONUNBUFFERED="1")
    env.setdefault("OMP_NUM_THREADS", "8")
    for attempt in (1, 2):
        # (the port is free when free_port() looks and may be taken when the rendezvous store binds it a second later: the ONE
        # failure that is tried again — with another port, announced on stderr — and only if no rank has printed anything yet)
        rc, records, port_taken = self_launch_once(args, argv, env, threading)
        if rc == 0 or not port_taken or records or attempt == 2:
            break
        print("bench.py: the rendezvous port was taken between probing and binding (EADDRINUSE); starting the ranks once more on another port",
              file=sys.stderr, flush=True)
    if rc != 0:
        raise SystemExit("bench.py --gpus %d: the ranks started with torch.distributed.run exited with code %d (their messages are above)"
                         % (args.gpus, rc))
    if not records:
        raise SystemExit("bench.py --gpus %d: rank 0 printed no result line" % args.gpus)


def self_launch_once(args, argv, env, threading):
    """one start of the ranks: (exit code, record lines relayed, did the rendezvous fail on a taken port)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1, env=env, start_new_session=True)
    records, seen = [], {"eaddrinuse": False}

    def relay():
        for ln in proc.stdout:
            ln = ln.rstrip("\n")
            if ln.startswith('{"metric"'):
                records.append(ln)
                print(ln, flush=True)
            else:
                if "EADDRINUSE" in ln:
                    seen["eaddrinuse"] = True
                print(ln, file=sys.stderr, flush=True)

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    try:
        proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, 9)     # exactly the process group started above
        proc.wait()
        th.join(timeout=10)
        raise SystemExit("bench.py --gpus %d: the ranks did not finish within %d s (%d record line(s) had been printed by then)"
                         % (args.gpus, args.launch_timeout, len(records)))
    th.join(timeout=30)
    return proc.returncode, records, seen["eaddrinuse"]


def roofline_of(nx, rows_local, steps, loop_ms, multistep, fused, deep, twin):
    """the roofline object of the module docstring for the dominant kernel of one leg"""
    steps_per_launch = multistep if multistep else (fused if fused else 1)
    launches = steps // steps_per_launch + steps % steps_per_launch
    if deep and not multistep and steps >= 2:
        # d2q9_deep: the run is split into the fewest launches, of equal depth (20 steps = 7+7+6)
        launches = -(-steps // fused)
        steps_per_launch = steps / launches
    launch_s = loop_ms * 1e-3 / launches
    cells_local = nx * rows_local
    model_bytes = (BYTES_PER_LU + MASK_BYTES) * cells_local
    achieved = model_bytes / launch_s / 1e9
    alg_gbps = BYTES_PER_LU * cells_local * steps_per_launch / launch_s / 1e9
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
        "kernel": ("d2q9_multi (%d timesteps per launch on LDS tiles)" % multistep) if multistep else (
            "d2q9_deep%s (up to %d timesteps per launch, lanes of two cells)" % ("_twin" if twin else "", fused) if deep else KERNELS[fused]),
        "launch_us": round(launch_s * 1e6, 2), "steps_per_launch": round(steps_per_launch, 3),
        "model_bytes_per_launch": model_bytes,
        "formula": "achieved = model_bytes_per_launch / launch_us; model_bytes_per_launch = (72 + 1) B x %d cells of the "
                   "rank's slab: a launch reads the grid once, writes it once and reads the byte mask, however many "
                   "timesteps it advances; launch_us = HIP-event time of the step loop / launches" % cells_local,
        "algorithmic": {"bytes_per_lattice_update": BYTES_PER_LU, "gbps": round(alg_gbps, 1),
                        "frac_of_peak": round(alg_gbps / HBM_PEAK_GBPS, 4),
                        "formula": "72 B x cells x steps_per_launch / launch_us (SURVEY.md 8d); above the 8 TB/s peak where "
                                   "temporal blocking keeps steps_per_launch-1 intermediate states on the chip: the "
                                   "speed-up over a perfect one-step-per-launch kernel, not a bandwidth"}}


OPTION_KEYS = ("fuse", "pair", "launch_steps", "multistep", "halo_depth", "resident")


def kernel_label(o):
    """name of the kernel a context runs, from its read-back options (OPTION_KEYS)"""
    if o.get("resident"):
        return "d2q9_resident (bands of %d rows held in registers over all timesteps of a launch)" % o["resident"]
    if o["multistep"]:
        return "d2q9_multi"
    if o["fuse"] >= 5:   # 5: the five-step chunk pairs of row slabs of 300K to 3M cells (compact launch sets)
        return "d2q9_deep%s" % ("_twin" if o["pair"] else "")
    return KERNELS[{0: 0, 1: 2, 3: 3, 4: 4}[o["fuse"]]]


def kernel_shape(opts):
    """(fused, deep, twin, multistep) from a context's read-back options"""
    fused = {0: 0, 1: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 7, 8: 8}[opts["fuse"]]   # timesteps per launch of the dominant kernel (0: one)
    deep = fused >= 5
    twin = bool(deep and opts["pair"])   # d2q9_deep_twin (chunk pairs: up to five steps per launch below 3M cells, up to eight from there)
    if deep:
        fused = opts["launch_steps"]
    return fused, deep, twin, opts["multistep"]


def attach_traffic(rf, lups, nx, ny, deep, twin, steps_per_launch, model_bytes, launch_s, valu_tera, version):
    """measured PMC traffic of the same command, when a profile of this workload AND of this build is committed"""
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tp):
        return
    with open(tp) as f:
        tj = json.load(f)
    key = "%dx%d/%s" % (nx, ny, ("deep_twin" if twin else "deep") if deep else "step%d" % steps_per_launch)
    if key not in tj:
        return
    ent = tj[key]
    if ent.get("library_version") != version:
        # counters of another build say nothing about this one: refuse them rather than let them go stale silently
        rf["traffic_refused"] = {"reason": "profiles/traffic.json[%s] was measured on library '%s', this is '%s'"
                                           % (key, ent.get("library_version"), version), "source": ent.get("source")}
        return
    tb = ent["hbm_bytes_per_launch"]
    rf["traffic"] = tb
    rf["traffic_measured_in_run"] = False
    rf["traffic_source"] = ent.get("source")
    # The bytes were counted on launches of ONE depth (the profile's own run: steps_per_launch, launch_us); this run may have
    # cut its steps into launches of other depths (20 steps = 7 + 7 + 6).  Price the bytes on the duration of the launches
    # they were counted on; without that record, on this run's launches only if they have the profile's depth.
    spl, lus = ent.get("steps_per_launch"), ent.get("launch_us")
    if lus:
        rf["traffic_frac"] = round(tb / (lus * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        rf["traffic_launch"] = {"steps_per_launch": spl, "launch_us": lus, "what": "the profiled launches the bytes were counted on"}
    elif spl is None or abs(rf["steps_per_launch"] - spl) < 1e-9:
        rf["traffic_frac"] = round(tb / launch_s / 1e9 / HBM_PEAK_GBPS, 4)
    else:
        rf["traffic_frac"] = None
    rf["traffic_over_model"] = round(tb / model_bytes, 4)
    if ent.get("evidence"):
        rf["bound_evidence"] = ent["evidence"]
        # `bound` names the roofline `frac` is priced against (the metric's: HBM); which resource the
        # kernel actually runs out of first is read off the counters of the committed profile
        vs = ent["evidence"].get("valu_issue_share")
        if vs is not None and rf["traffic_frac"] is not None:
            rf["limited_by"] = "valu_issue" if vs > rf["traffic_frac"] else "hbm"
    if valu_tera:
        # issue-rate roofline: VALU lane-instructions the kernel executes per lattice update (committed profile) x rate
        lane = (rf.get("bound_evidence") or {}).get("valu_lane_instr_per_cell_step")
        rf["valu"] = {"issue_rate_measured": valu_tera, "unit": "1e12 packed-fp32 lane-instructions/s", "theoretical": 39.3,
                      "lane_instr_per_cell_step": lane, "lane_instr_source": "SQ_INSTS_VALU of the committed profile" if lane else None,
                      "achieved": round(lane * lups / 1e12, 2) if lane else None,
                      "frac": round(lane * lups / 1e12 / valu_tera, 4) if lane else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--nx", type=int, default=8192)
    ap.add_argument("--ny", type=int, default=8192)
    ap.add_argument("--workload", default="cavity", choices=["cavity", "empty", "tiled"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="what `value` is: strong = the nx x ny grid split over the ranks (default; the weak-scaling figure is "
                         "then the `weak` object of the same line), weak = every rank gets ny rows")
    ap.add_argument("--accel", type=float, default=0.005)
    ap.add_argument("--valu-calib", type=int, default=40, help="launches (~2 ms each) of the packed-FMA issue-rate calibration; 0 = skip")
    ap.add_argument("--calib-iters", type=int, default=10, help="launches of the 1 GiB copy kernel that measures the roofline denominator")
    ap.add_argument("--fuse", type=int, default=-1, help="timesteps per launch of the register/LDS-window kernels: 0, 1 (= 2), 3, 4, 6..8 (d2q9_deep, at most); -1: library default")
    ap.add_argument("--transport", default="both", choices=["both", "peer", "rccl"],
                    help="N > 1: halo transport(s) to check against the oracle and measure; the faster one that passed is `value`")
    ap.add_argument("--launcher", default="auto", choices=["auto", "torchrun", "one-process"],
                    help="--gpus N without WORLD_SIZE in the environment: auto/torchrun = start `python -m torch.distributed.run` "
                         "with N ranks as a child process (torchrun also for N = 1: the rank path on one GPU); one-process = "
                         "one process drives N slabs (lbm_create(ndev=N))")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="seconds the self-launched ranks get")
    ap.add_argument("--budget-s", type=float, default=600.0,
                    help="N > 1: a side leg (1024x1024 strong scaling, weak scaling) is started only while the run is younger than this "
                         "many seconds; the headline record is printed before them either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip `value_cold` (the same steps before any calibration launch)")
    ap.add_argument("--no-extra", action="store_true", help="skip the 1024x1024, weak-scaling and reference-rule side measurements")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    one_process = False
    if not launched and (args.gpus > 1 or args.launcher == "torchrun"):
        import importlib.util
        have = importlib.util.find_spec("torch") is not None and importlib.util.find_spec("torch.distributed.run") is not None
        if args.launcher != "one-process" and have:
            return self_launch(args, sys.argv[1:])
        one_process = True   # no launcher on this machine (or asked for): INTEGRATION.md section 3's one-process form

    # (the host driver supports dmabuf IPC only: without this RCCL and HIP IPC between ranks fail with hipIpcGetMemHandle:
    # invalid argument; the GPU boxes export it already — a rank started by somebody else's launcher must not depend on that)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # device plumbing + torch.distributed (RCCL) only
    import lbm_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    visible = torch.cuda.device_count()
    rehearsal = os.environ.get("LBM_BENCH_REHEARSAL") == "1"
    if max(world, args.gpus) > visible and not rehearsal:
        if rank == 0:
            print("bench.py: %d GPUs needed, %d visible" % (max(world, args.gpus), visible), file=sys.stderr, flush=True)
        raise SystemExit(2)
    local_rank = local_rank % visible     # (only a rehearsal has more ranks than devices)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # A single rank started by the launcher (LBM_BENCH_CHILD, or LBM_BENCH_RANK_MODE=1 by hand) drives the
    # one-process-per-GPU code path as a ring of one: the rank is its own ring neighbour (default "force_halo"), so
    # torch.distributed + the library's RCCL communicator + both transports run on a one-GPU box
    rank_mode = world > 1 or os.environ.get("LBM_BENCH_RANK_MODE") == "1" or (launched and os.environ.get("LBM_BENCH_CHILD") == "1")
    if rank_mode and world == 1:
        lbm_amd.set_default("force_halo", 1)
    if rehearsal and not rank_mode and not one_process:
        raise SystemExit("LBM_BENCH_REHEARSAL=1 needs ranks or slabs: --gpus N with N > 1, or --launcher torchrun")
    dist = init_dist("gloo" if rehearsal else "nccl", rank, world, device) if rank_mode else None
    coll_device = torch.device("cpu") if rehearsal else device   # where the collectives' tensors live
    env = Env(lbm_amd, torch, dist, rank, world, local_rank, coll_device, rehearsal)
    version = lbm_amd.load_library().lbm_version().decode()

    nx = args.nx
    ny = args.ny * (world if args.scaling == "weak" else 1)
    total_steps = args.warmup + args.steps + 256
    obstacles = make_workload(args.workload, nx, ny)
    params = lbm_amd.make_params(nx, ny, total_steps, 10, 0.1, args.accel, 1.85, obstacles)
    ndev = args.gpus if one_process else 1

    def plain_leg(p, ob, warmup, steps, fuse=-1, want_av=False, before=None):
        """one context in this process (one slab, or --launcher one-process: one slab per device)"""
        with (lbm_amd.LBM(p, ob, devices=[i % visible for i in range(ndev)]) if ndev > 1 else lbm_amd.LBM(p, ob)) as sim:
            if fuse >= 0:
                sim.set_option("fuse", fuse)
            sim.upload(None)
            if before:
                before()
            e = {}
            e["wall_s"], e["loop_ms"] = timed_run(env, sim, warmup, steps)
            e["row_range"] = sim.row_range()
            e["options"] = {k: sim.get_option(k) for k in OPTION_KEYS}
            if want_av:
                e["av"] = sim.download(cells=False)[1]
            if ndev > 1:
                per = max(e["options"]["multistep"], e["options"]["launch_steps"], 1)
                st = sim.run_profiled(8 * per)
                e["per_rank"] = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}]
            return e

    # ---- `value_cold`: the same W + K steps BEFORE anything else has run on the chip (no copy calibration, no issue-rate
    # calibration): the first ~40 ms of the VALU-bound default kernel run 8-10 % below its steady state on a chip that has
    # not been loaded yet (profiles/r02_cold_start.txt), and a 20-step record is mostly those milliseconds
    cold = None
    if not args.no_cold:
        if rank_mode:
            c = list(rank_leg(env, params, obstacles, ["peer" if rehearsal else "rccl"], args.warmup, args.steps, args.fuse,
                              profile=False).values())[0]
        else:
            c = plain_leg(params, obstacles, args.warmup, args.steps, args.fuse)
        if "wall_s" in c:
            cold = round(nx * ny * args.steps / c["wall_s"] / 1e6, 1)

    # ---- the two roofline denominators (a float4 copy of 1 GiB each way; packed-FMA issue rate, ~1.9 ms per launch) are
    # measured anyway — and measured BETWEEN the creation of a timed leg's context and its W warm-up steps they also bring
    # the chip to its working clock (context creation leaves it idle for a second; the first ~40 ms after that run 8-25 %
    # below the steady state).  That is a warm-up beyond --warmup, so the line says so: `pre_warmup`, and `value_cold`.
    calib = {"copy_gbps": None, "valu_tera": None}
    pre = {"copy_launches": 0, "copy_ms": 0.0, "valu_calib_launches": 0, "valu_calib_ms": 0.0, "legs": 0,
           "what": "calibration launches (1 GiB float4 copy: once; packed-FMA issue rate: before every timed leg) that run "
                   "between the creation of a leg's context and its W warm-up steps; `value_cold` is the same W + K steps "
                   "with none of them, in a context of its own, measured first"}

    def pre_warm():
        try:
            if calib["copy_gbps"] is None and args.calib_iters > 0:
                t0 = time.perf_counter()
                calib["copy_gbps"] = round(lbm_amd.copy_bandwidth_gbps(1 << 30, args.calib_iters), 1)
                pre["copy_launches"], pre["copy_ms"] = args.calib_iters, round((time.perf_counter() - t0) * 1e3, 1)
            if args.valu_calib > 0:
                t0 = time.perf_counter()
                calib["valu_tera"] = round(lbm_amd.valu_rate_tera(args.valu_calib), 2)
                pre["valu_calib_launches"], pre["valu_calib_ms"] = args.valu_calib, round((time.perf_counter() - t0) * 1e3, 1)
                pre["legs"] += 1
        except lbm_amd.LBMError:
            pass

    # ---- N ranks: every transport against the oracle first; only those that pass are timed ---------------------------
    tcheck, transports = None, [None]
    if rank_mode:
        want = ["peer"] if rehearsal else {"both": ["rccl", "peer"], "peer": ["peer"], "rccl": ["rccl"]}[args.transport]
        log("checking transports %s on %d rank(s) before anything is timed" % (want, world), rank)
        tcheck = transport_check(env, params, obstacles, want)
        transports = [t for t in want if tcheck["transports"][t]["ok"]]
        if "rccl" in want and "rccl" not in transports:
            # RCCL's launch sets are "staged" since round 4 (one launch per set, the exchange behind hipStreamWaitValue32): a form that,
            # like everything here, has met one device only.  Should it fail on the machine at hand, the two-stream sets of round 3
            # (option "compact" 0) get the same three checks and take its place.
            log("rccl (staged launch sets) failed %s: checking its two-stream form (rccl2)" % tcheck["transports"]["rccl"].get("failed_legs"), rank)
            t2 = transport_check(env, params, obstacles, ["rccl2"])
            tcheck["transports"]["rccl2"] = t2["transports"]["rccl2"]
            for k in t2["legs"]:
                tcheck["legs"][k]["transports"]["rccl2"] = t2["legs"][k]["transports"]["rccl2"]
            if t2["transports"]["rccl2"]["ok"]:
                transports.append("rccl2")
        log("transports that passed: %s" % transports, rank)

    # ---- the timed region(s): one per halo transport, the faster one is reported as `value` -----------------------
    if rank_mode:
        runs = rank_leg(env, params, obstacles, transports, args.warmup, args.steps, args.fuse, want_av=True, before=pre_warm)
    else:
        runs = {"single" if ndev == 1 else "one-process": plain_leg(params, obstacles, args.warmup, args.steps, args.fuse, want_av=True,
                                                                    before=pre_warm)}
    copy_gbps, valu_tera = calib["copy_gbps"], calib["valu_tera"]
    good = {k: v for k, v in runs.items() if "wall_s" in v}

    out, ok = None, False
    av_oracle = []
    if rank == 0 and not good:
        out = {"metric": "MLUPS", "value": None, "unit": "MLUPS (million lattice updates/s)", "n_gpus": max(world, ndev),
               "steps": args.steps, "warmup": args.warmup, "result_ok": False, "transport_check": tcheck,
               "transports": {k: v for k, v in runs.items()},
               "error": "no halo transport passed the oracle check and completed the timed run"}
    if rank == 0 and good:
        best = min(good, key=lambda k: good[k]["wall_s"])
        b = good[best]
        wall, loop_ms = b["wall_s"], b["loop_ms"]
        y0, y1 = b["row_range"]
        fused, deep, twin, multistep = kernel_shape(b["options"])
        lups = nx * ny * args.steps / wall
        rows_local = (y1 - y0) if ndev == 1 else ny // ndev
        rf = roofline_of(nx, rows_local, args.steps, loop_ms, multistep, fused, deep, twin)
        out = {
            "metric": "MLUPS", "value": round(lups / 1e6, 1), "unit": "MLUPS (million lattice updates/s)",
            "n_gpus": max(world, ndev), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 5), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d %s, D2Q9-BGK fused timestep, uniform rest start" % (nx, ny, {
                "cavity": "lid-driven cavity (4 border lines blocked)", "empty": "no obstacles (periodic)",
                "tiled": "obstacles_1024x1024 tiled up"}[args.workload]),
                "nx": nx, "ny": ny, "rows_per_gpu": rows_local, "partition": "rows x%d" % max(world, ndev),
                "omega": 1.85, "accel": args.accel, "density": 0.1},
            "roofline": rf, "library": version,
            "launcher": ("torch.distributed.run started by bench.py itself" if os.environ.get("LBM_BENCH_CHILD") == "1" else
                         "external (WORLD_SIZE in the environment)") if launched else
                        ("one process, lbm_create(ndev=%d)" % ndev if ndev > 1 else "none (single GPU)"),
            "value_cold": cold, "pre_warmup": pre,
        }
        if world == 1 and ndev == 1 and not multistep:
            attach_traffic(rf, lups, nx, ny, deep, twin, int(round(rf["steps_per_launch"])) if not deep else 0,
                           rf["model_bytes_per_launch"], rf["launch_us"] * 1e-6, valu_tera, version)
        if valu_tera and "valu" not in rf:
            rf["valu"] = {"issue_rate_measured": valu_tera, "unit": "1e12 packed-fp32 lane-instructions/s", "theoretical": 39.3,
                          "lane_instr_per_cell_step": None, "achieved": None, "frac": None}
        # what a plain float4 copy achieves on this box right now (context for `frac`; the spec peak stays `peak`)
        if copy_gbps:
            rf["copy_kernel_gbps"] = copy_gbps
            rf["frac_of_copy_kernel"] = round(rf["achieved"] / copy_gbps, 4)
        if rank_mode:
            out["transports"] = {k: ({"value": round(nx * ny * args.steps / v["wall_s"] / 1e6, 1),
                                      "ms_per_step": round(v["wall_s"] * 1e3 / args.steps, 5)} if "wall_s" in v else v)
                                 for k, v in runs.items()}
            out["transport"] = best
            out["transport_check"] = tcheck
            out["rccl_world_size"] = 0 if rehearsal else world
            if rehearsal:
                out["rehearsal"] = ("LBM_BENCH_REHEARSAL=1: %d ranks as processes on %d GPU(s), gloo rendezvous, no RCCL communicator, peer "
                                    "stores over HIP IPC — a test of the N > 1 code path, NOT a benchmark" % (world, visible))
        if b.get("per_rank"):
            out["per_rank_launch_set_us"] = b["per_rank"]

    def judge_result():
        """`result_ok`: av_vels of the timed context against the oracle's record of the same steps"""
        if not av_oracle:
            # no baseline leg in this run: a few steps of the OpenMP build of the same oracle, for the check alone
            os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, len(os.sched_getaffinity(0)) // max(1, world))))
            oracle_rate("f32", nx, ny, obstacles, args.accel, 6.0, 8, omp=True, av_out=av_oracle)
        good_ok, detail = result_check(good[best]["av"], np.array(av_oracle))
        out["result_ok"] = good_ok
        out["result_check"] = detail
        return good_ok

    if rank == 0 and out is not None and good and rank_mode:
        # ---- N ranks: the headline record NOW, before the side legs — a run that is cut off later has left it behind
        ok = judge_result()
        print(json.dumps(dict(out, partial="headline record; the full record (side legs `also`, `weak`) is the last line of this run")),
              flush=True)
        log("headline record printed: %s MLUPS over %s, result_ok %s" % (out["value"], out.get("transport"), ok), rank)

    def side_leg_allowed(name):
        """the ranks' common decision (a collective) whether a side leg still starts within --budget-s"""
        late = max_over_ranks(dist, [time.time() - T_START], coll_device)[0] > args.budget_s
        if late and rank == 0 and out is not None:
            out.setdefault("skipped_legs", []).append("%s: the run was older than --budget-s %.0f s" % (name, args.budget_s))
        return not late

    extra = not args.no_extra and (nx, ny) != (1024, 1024)
    if extra and rank_mode and transports:
        if side_leg_allowed("also"):
            # ---- the 1024x1024 input of the reference row-partitioned over the same ranks (BASELINE config 4: strong scaling)
            p2, ob2 = shipped("1024x1024")
            n2, w2 = 4000, 400
            p2.max_iters = n2 + w2 + 256
            r2 = rank_leg(env, p2, ob2, transports, w2, n2, before=pre_warm)
            g2 = {k: v for k, v in r2.items() if "wall_s" in v}
            if rank == 0 and out is not None and g2:
                b2 = min(g2, key=lambda k: g2[k]["wall_s"])
                out["also"] = {"workload": "input_1024x1024.params + obstacles_1024x1024.dat, rows x%d (strong scaling of the reference's "
                                           "largest input)" % world,
                               "value": round(1024 * 1024 * n2 / g2[b2]["wall_s"] / 1e6, 1), "unit": "MLUPS", "steps": n2, "warmup": w2, "transport": b2,
                               "transports": {k: (round(1024 * 1024 * n2 / v["wall_s"] / 1e6, 1) if "wall_s" in v else v) for k, v in r2.items()},
                               "us_per_step": round(g2[b2]["wall_s"] / n2 * 1e6, 3), "halo_depth": g2[b2]["options"]["halo_depth"],
                               "per_rank_launch_set_us": g2[b2].get("per_rank")}
        # ---- BASELINE config 5's weak-scaling leg: every rank holds args.ny rows of an nx x (ny * N) cavity
        if args.scaling == "strong" and side_leg_allowed("weak"):
            nyw = args.ny * world
            obw = make_workload(args.workload, nx, nyw)
            nw, ww = min(args.steps, 96), min(args.warmup, 16)
            pw = lbm_amd.make_params(nx, nyw, nw + ww + 256, 10, 0.1, args.accel, 1.85, obw)
            rw = rank_leg(env, pw, obw, transports, ww, nw, args.fuse, before=pre_warm)
            del obw
            gw = {k: v for k, v in rw.items() if "wall_s" in v}
            if rank == 0 and out is not None and gw:
                bw = min(gw, key=lambda k: gw[k]["wall_s"])
                out["weak"] = {"workload": "%dx%d %s: %d rows per rank" % (nx, nyw, args.workload, args.ny), "scaling": "weak",
                               "value": round(nx * nyw * nw / gw[bw]["wall_s"] / 1e6, 1), "unit": "MLUPS", "steps": nw, "warmup": ww,
                               "ms_per_step": round(gw[bw]["wall_s"] * 1e3 / nw, 5), "transport": bw,
                               "per_gpu": round(nx * nyw * nw / gw[bw]["wall_s"] / 1e6 / world, 1),
                               "transports": {k: (round(nx * nyw * nw / v["wall_s"] / 1e6, 1) if "wall_s" in v else v) for k, v in rw.items()},
                               "per_rank_launch_set_us": gw[bw].get("per_rank")}
    if world == 1 and rank == 0 and not rank_mode and out is not None and good:
        if extra and ndev == 1:
            # the reference's own largest input, for the 1024x1024 figure the north star asks for
            p2, ob2 = shipped("1024x1024")
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
                f2 = s2.get_option("launch_steps")
                k2 = kernel_label({k: s2.get_option(k) for k in OPTION_KEYS})
            note = ("the grid is read and written once per launch of up to 256 timesteps and held in registers in between; what moves per step is "
                    "13 MB of exchange rows, served on-die — profiles/r04c3.txt" if "resident" in k2 else
                    "both grids fit the 256 MiB Infinity Cache: these bytes come from cache, not HBM — profiles/r02_config3.txt")
            out["also"] = {"workload": "input_1024x1024.params + obstacles_1024x1024.dat (%s)" % note,
                           "value": round(1024 * 1024 * n2 / w2 / 1e6, 1), "unit": "MLUPS", "steps": n2, "us_per_step": round(ms2 / n2 * 1e3, 3),
                           "steps_per_launch": f2, "kernel": k2,
                           "model_gbps": round((BYTES_PER_LU + MASK_BYTES) * 1024 * 1024 / (ms2 * 1e-3 / n2 * f2) / 1e9, 1),
                           "algorithmic_gbps": round(BYTES_PER_LU * 1024 * 1024 / (ms2 * 1e-3 / n2) / 1e9, 1)}
            # the HBM-bound members of the kernel family on the same grid, same run: what `frac` looks like for a kernel that IS
            # bound by the roofline the metric names (the default above trades that for fewer passes over the grid)
            if deep and (nx, ny) == (8192, 8192):
                sib = {}
                for f4, label in ((4, "d2q9_step4"), (0, "d2q9_step")):
                    per = max(f4, 1)
                    n4 = 96 if f4 else 24
                    p4 = lbm_amd.make_params(nx, ny, n4 + 48, 10, 0.1, args.accel, 1.85, obstacles)
                    with lbm_amd.LBM(p4, obstacles) as s4:
                        s4.set_option("fuse", f4)
                        s4.upload(None)
                        s4.run(48)
                        ms4 = s4.run_timed(n4)
                    l4 = ms4 * 1e-3 / (n4 // per)
                    sib[label] = {"value": round(nx * ny * n4 / ms4 / 1e3, 1), "unit": "MLUPS", "steps_per_launch": per,
                                  "launch_us": round(l4 * 1e6, 2),
                                  "frac": round((BYTES_PER_LU + MASK_BYTES) * nx * ny / l4 / 1e9 / HBM_PEAK_GBPS, 4)}
                out["roofline"]["hbm_bound_siblings"] = sib
            # reference-rule figures (d2q9-bgk.c:196-263: initial state + step loop + read-back of av_vels and the state)
            ref = {}
            p3 = lbm_amd.make_params(nx, ny, args.steps, 10, 0.1, args.accel, 1.85, obstacles)
            with lbm_amd.LBM(p3, obstacles) as s3:
                # the read-back targets exist before the clock starts, as in the reference (malloc in initialise(),
                # d2q9-bgk.c:519-526) — page-locked (lbm_host_alloc), which is what host/d2q9-bgk.c does too
                hb = lbm_amd.HostBuffer((4, ny, nx))
                t1 = time.perf_counter()
                s3.upload(None)
                s3.run(args.steps)
                s3.sync()
                s3.download(cells=False)
                s3.final_state(out=hb.array)
                s3.reynolds()
                tr = time.perf_counter() - t1
                hb.close()
            ref["headline_grid"] = {"value": round(nx * ny * args.steps / tr / 1e6, 1), "unit": "MLUPS", "steps": args.steps,
                                    "elapsed_s": round(tr, 4),
                                    "what": "device-side initial state + %d steps + av_vels + the four final_state columns "
                                            "(%.2f GB to page-locked host memory, lbm_host_alloc) + Reynolds number, wall clock" % (args.steps, 4 * nx * ny * 4 / 1e9)}
            try:
                with tempfile.TemporaryDirectory() as d:
                    r = subprocess.run([os.path.join(ROOT, "d2q9-bgk.exe"), os.path.join(ROOT, "inputs", "input_1024x1024.params"),
                                        os.path.join(ROOT, "inputs", "obstacles_1024x1024.dat")], cwd=d, capture_output=True, text=True,
                                       timeout=300, env=dict(os.environ, LBM_NO_OUTPUT="1"))
                el = float([ln for ln in r.stdout.splitlines() if ln.startswith("Elapsed time:")][0].split()[2])
                ref["c_host_input_1024x1024"] = {"value": round(1024 * 1024 * 20000 / el / 1e6, 1), "unit": "MLUPS", "steps": 20000,
                                                 "elapsed_s": el, "what": "./d2q9-bgk.exe on the shipped input: its 'Elapsed time' line"}
            except Exception as e:
                ref["c_host_input_1024x1024"] = {"error": str(e)[:200]}
            out["reference_rule"] = ref
    if rank == 0 and out is not None and good:
        # ---- CPU baseline (rank 0, N = 1 only) and the oracle record `result_ok` is judged on -------------------------
        if world == 1 and ndev == 1 and not args.no_cpu_baseline and not rank_mode:
            out["cpu_baseline"] = cpu_baseline(nx, ny, obstacles, args.accel, av_out=av_oracle)
        if "result_ok" not in out:
            ok = judge_result()
    if rank == 0:
        print(json.dumps(out), flush=True)
        if not ok:
            print("bench.py: result_ok is false: %s" % json.dumps({k: (out or {}).get(k) for k in ("result_check", "error", "transports", "transport_check")}),
                  file=sys.stderr, flush=True)
    if dist is not None:
        # the line is out; a rendezvous backend that objects to the order in which the ranks hang up must not turn a good
        # record into a failed run
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            print("bench.py: rank %d: tear-down of the process group failed (ignored): %s" % (rank, str(e)[:300]), file=sys.stderr, flush=True)
    if rank == 0 and not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()

"""Helper of test_gpu_parity.py::test_transport_self_ring — run as a subprocess (so that a hang in the
communication layer is a test failure, not a stuck test session).  One rank with the default "force_halo": the
slab is its own north and south neighbour and every halo exchange goes through the chosen transport to itself:
  rccl   ncclSend/ncclRecv on a communicator made by ncclCommInitRank, av_vels through ncclAllReduce
  copy   device-to-device hipMemcpyAsync between the slab's own rows
  peer   halo_push kernel + flag words, consumer side by wait kernel (halo_sync 0) and by hipStreamWaitValue32 (1);
         once on a context that also has a communicator (lbm_connect_peers on top of RCCL, switched back and forth)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd

transport = sys.argv[1]  # "rccl", "copy" or "peer"; "rccl_deep": the staged launch sets of a slab big enough for the deep window kernel
rng = np.random.default_rng(21)
if transport == "rccl_deep":
    # RCCL + d2q9_deep: ONE launch per launch set (edge units first, interior chunk pairs, side walls balanced, free sweeps) whose edge
    # units store their rows into the slab's staging blocks; the edge stream waits on the flag word (hipStreamWaitValue32), sends the
    # blocks to itself and receives into the halo rows beside the running interior.  23 steps = 8 + 8 + 7; the two-stream form
    # ("compact" 0); split runs; and the test hook: an exchange that delivers nothing must change the result
    # (second case: a slab of 1.2M cells — five halo rows, the five-step chunk pairs with their edge pairs, staged the same way)
    big = len(sys.argv) < 3 or sys.argv[2] != "twin5"
    nx, ny, nsteps = (8192, 416, 23) if big else (2048, 700, 23)
    want_fuse, want_halo = (8, 8) if big else (5, 5)
    ob = (rng.random((ny, nx)) < 0.0002).astype(np.int32)
    ob[0, :] = ob[-1, :] = 0
    ob[:, 0] = ob[:, -1] = 1
    ob[100:300, 64:-64] = 0
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
    cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
    p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.set_option("fuse", 0)
        sim.set_option("multistep", 0)
        sim.upload(cells0)
        sim.run(nsteps)
        ref, av_ref = sim.download()
    lbm_amd.set_default("force_halo", 1)
    lbm_amd.set_default("transport", "rccl")
    for (compact, pair, split, stale) in ((-1, -1, 0, 0), (-1, 0, 0, 0), (0, -1, 0, 0), (-1, -1, 9, 0), (-1, -1, 0, 2)):
        with lbm_amd.LBM(p, ob, rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) as sim:
            sim.set_option("compact", compact)
            sim.set_option("pair", pair)
            assert sim.get_option("transport") == 1 and sim.get_option("halo_depth") == want_halo
            if big:
                assert sim.get_option("fuse") == 8
                assert sim.get_option("compact") == (1 if compact else 0) and sim.get_option("pair") == (1 if compact and pair else 0)
            else:   # the five-step pairs exist as staged sets only: without "compact" or "pair" the four-step kernel on two streams
                on = bool(compact and pair)
                assert sim.get_option("fuse") == (5 if on else 4) and sim.get_option("compact") == (1 if on else 0)
            if stale:
                sim.set_option("debug_stale_exchange", stale)
            sim.upload(cells0)
            if split:
                sim.run(split)
                sim.sync()
            sim.run(nsteps - split)
            got, av = sim.download()
        tag = "rccl staged sets (%dx%d): compact %d pair %d split %d stale %d" % (nx, ny, compact, pair, split, stale)
        same = np.array_equal(got, ref)
        if stale:
            assert not same, tag + ": a lost exchange went unnoticed"
        else:
            if not same:
                bad = np.argwhere(np.any(got != ref, axis=(0, 2))).ravel()
                raise AssertionError("state differs (%s): %d rows, first %s, last %s" % (tag, bad.size, bad[:6], bad[-6:]))
            assert np.max(np.abs(av - av_ref) / av_ref) < 2e-6, tag
        print("ok:", tag, flush=True)
    print("self-ring ok:", transport)
    sys.exit(0)
nx, ny, nsteps = 512, 96, 23
ob = (rng.random((ny, nx)) < 0.05).astype(np.int32)
ob[0, :] = ob[-1, :] = 0
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
p = lbm_amd.make_params(nx, ny, nsteps, obstacles=ob)

with lbm_amd.LBM(p, ob) as sim:
    sim.set_option("fuse", 0)
    sim.set_option("multistep", 0)
    sim.upload(cells0)
    sim.run(nsteps)
    ref, av_ref = sim.download()
    re_ref = sim.reynolds()

lbm_amd.set_default("force_halo", 1)
lbm_amd.set_default("transport", transport)
code = lbm_amd.TRANSPORTS[transport]
# peer: consumer-side wait by kernel / by hipStreamWaitValue32 / inside the consuming kernel (compact sets), and the launch sets of d2q9_multi compact (one launch,
# the edge tiles push the halo rows themselves) / as edge launch + interior launch + push kernel
variants = ((0, -1), (1, -1), (2, -1), (0, 0)) if transport == "peer" else ((0, -1),)
for (sync, compact) in variants:
    # 1 / 2 / 3 / 4 / up to 8 and up to 6 (d2q9_deep) / 8 / 5 (d2q9_multi) timesteps per launch set (halo depth 8)
    for (fuse, ms) in ((0, 0), (1, 0), (3, 0), (4, 0), (8, 0), (6, 0), (0, 8), (0, 5)):
        kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) if transport == "rccl" else dict(devices=[0])
        with lbm_amd.LBM(p, ob, **kw) as sim:
            assert sim.get_option("transport") == code
            if transport == "peer":
                sim.set_option("halo_sync", sync)
                sim.set_option("compact", compact)
            sim.set_option("fuse", fuse)
            sim.set_option("multistep", ms)
            if transport == "peer":
                # compact launch sets exist for the LDS-tile kernel, the three- / four-step kernels and d2q9_deep (the slab is
                # small: halo depth 8, so fuse 4 and fuse 8 are allowed and used)
                assert sim.get_option("compact") == (1 if (compact and (ms or fuse >= 3)) else 0)
            assert ms or sim.get_option("fuse") == fuse
            sim.upload(cells0)
            sim.run(nsteps)
            got, av = sim.download()   # rank mode: av_vels go through ncclAllReduce
            re = sim.reynolds()
        tag = "transport %s, halo_sync %d, compact %d, fuse %d, multistep %d" % (transport, sync, compact, fuse, ms)
        if not np.array_equal(got, ref):
            bad = np.argwhere(np.any(got != ref, axis=(0, 2))).ravel()
            raise AssertionError("state differs (%s): %d rows, first %s, last %s" % (tag, bad.size, bad[:6], bad[-6:]))
        assert np.max(np.abs(av - av_ref) / av_ref) < 2e-6, tag
        assert abs(re / re_ref - 1) < 1e-5, tag
        print("ok:", tag, flush=True)

if transport == "peer":
    # a rank context with a communicator AND connected peers: starts on RCCL, connects to itself through its own
    # descriptor (same process: raw pointers), runs on peer stores, switches back to RCCL and forth again mid-run
    lbm_amd.set_default("transport", "auto")
    with lbm_amd.LBM(p, ob, rank=0, nranks=1, device=0, comm=lbm_amd.comm_id()) as sim:
        assert sim.get_option("transport") == 1
        info = sim.peer_info()
        sim.connect_peers(info, info)
        assert sim.get_option("transport") == 1   # a context with a communicator stays on RCCL until told otherwise
        sim.connect_peers(info, info)             # a repeated call replaces the links (nothing stays mapped twice)
        sim.set_option("transport", 3)
        assert sim.get_option("transport") == 3
        sim.upload(cells0)
        sim.run(7)
        sim.set_option("transport", 1)
        sim.run(9)
        sim.set_option("transport", 3)
        sim.run(nsteps - 16)
        got, av = sim.download()
    assert np.array_equal(got, ref) and np.max(np.abs(av - av_ref) / av_ref) < 2e-6
print("self-ring ok:", transport)

#!/usr/bin/env python3
"""tools/ab.py — the one A/B harness for kernel and schedule experiments (replaces the round-1 pile of
tools/ab_*.py one-offs; their measurements are quoted in DESIGN.md and live in the git history).

    python tools/ab.py --sizes 1024x1024,8192x8192 --opts "fuse=3;fuse=4,pair=0;fuse=4,pair=1"
    python tools/ab.py --sizes 8192x1024 --ring rccl,peer --opts ";fuse=3"         # a rank's slab, ring of one
    python tools/ab.py --sizes 2048x2048 --libs liblbm_hip_base.so,liblbm_hip.so  # two builds, interleaved
    python tools/ab.py --sizes 1024x300 --opts "fuse=4,chunk_rows=16" --check     # + bit identity vs single steps

Every (size, library, ring mode, option set) is timed with lbm_run_timed: best of --reps runs of --steps steps
(default: enough steps for ~1e9 lattice updates), after 48 warm-up steps.  Option sets are `;`-separated lists of
key=value pairs for lbm_set_option ("" = library defaults); halo_depth / lanes_out / transport / force_halo are
creation defaults (lbm_set_default) instead.  --libs re-runs this script once per library in a child process (LBM_LIB is read when the binding is
loaded), rounds interleaved so that clock drift of the box hits all builds alike.
Workload: cavity (4 border lines blocked) unless --workload empty|tiled|random."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def obstacles(kind, nx, ny, rng):
    ob = np.zeros((ny, nx), np.int32)
    if kind == "cavity":
        ob[0, :] = ob[-1, :] = 1
        ob[:, 0] = ob[:, -1] = 1
    elif kind == "walls":      # side walls only: what a rank's slab of a cavity looks like
        ob[:, 0] = ob[:, -1] = 1
    elif kind == "random":
        ob = (rng.random((ny, nx)) < 0.05).astype(np.int32)
    elif kind == "tiled":
        import lbm_amd
        _, ob1 = lbm_amd.read_inputs(os.path.join(ROOT, "inputs", "input_1024x1024.params"),
                                     os.path.join(ROOT, "inputs", "obstacles_1024x1024.dat"))
        ob = np.tile(ob1, (max(1, ny // 1024), max(1, nx // 1024)))[:ny, :nx].copy()
    return ob


DEFAULT_KEYS = ("force_halo", "halo_depth", "transport", "lanes_out")   # lbm_set_default keys; the rest are lbm_set_option keys


def parse_opts(text):
    opts, defaults = {}, {}
    for kv in filter(None, (t.strip() for t in text.split(","))):
        k, v = kv.split("=")
        (defaults if k in DEFAULT_KEYS else opts)[k] = int(v)
    return opts, defaults


def run_case(args, nx, ny, ring, optset):
    import lbm_amd
    opts, defaults = parse_opts(optset)
    if ring:
        defaults = dict(defaults, force_halo=1, transport=lbm_amd.TRANSPORTS[ring])
    for k, v in defaults.items():
        lbm_amd.set_default(k, v)
    try:
        rng = np.random.default_rng(1)
        ob = obstacles(args.workload if not ring or args.workload != "cavity" else "walls", nx, ny, rng)
        steps = args.steps or int(min(20000, max(96, 1.0e9 / (nx * ny)))) // 24 * 24
        p = lbm_amd.make_params(nx, ny, steps * args.reps + 64, obstacles=ob)
        kw = {}
        if ring == "rccl":
            kw = dict(rank=0, nranks=1, device=0, comm=lbm_amd.comm_id())
        elif ring:
            kw = dict(devices=[0])
        elif args.slabs > 1:
            kw = dict(devices=[0] * args.slabs)
        with lbm_amd.LBM(p, ob, **kw) as sim:
            for k, v in opts.items():
                sim.set_option(k, v)
            sim.upload(None)
            sim.run(48)
            ms = min(sim.run_timed(steps) for _ in range(args.reps))
            eff = {k: sim.get_option(k) for k in ("fuse", "multistep", "pair")}
            res = {"us_per_step": ms / steps * 1e3, "mlups": nx * ny * steps / ms / 1e3, "steps": steps, "effective": eff}
        if args.check:
            n = 13
            w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float64).reshape(9, 1, 1) * 0.1
            cells0 = (w * (1.0 + 0.2 * (rng.random((9, ny, nx)) - 0.5))).astype(np.float32)
            obr = obstacles("random", nx, ny, rng)
            pc = lbm_amd.make_params(nx, ny, n, obstacles=obr)
            outs = []
            for o, k2 in (({"fuse": 0, "multistep": 0}, {}), (opts, kw)):
                with lbm_amd.LBM(pc, obr, **k2) as sim:
                    for k, v in o.items():
                        sim.set_option(k, v)
                    sim.upload(cells0)
                    sim.run(n)
                    outs.append(sim.download())
            res["identical"] = bool(np.array_equal(outs[0][0], outs[1][0]))
            res["av_rel"] = float(np.max(np.abs(outs[0][1] - outs[1][1]) / np.abs(outs[0][1])))
        return res
    finally:
        for k in ("force_halo", "halo_depth", "transport", "lanes_out"):
            lbm_amd.set_default(k, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024x1024,8192x8192")
    ap.add_argument("--opts", default="", help="';'-separated option sets, each 'key=value,key=value'")
    ap.add_argument("--ring", default="", help="comma list of ring-of-one transports (rccl, peer, copy); empty = one plain slab")
    ap.add_argument("--slabs", type=int, default=1, help="several slabs on device 0 (one process)")
    ap.add_argument("--libs", default="", help="comma list of builds under the package directory")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2, help="with --libs: interleaved repetitions")
    ap.add_argument("--workload", default="cavity", choices=["cavity", "empty", "tiled", "random", "walls"])
    ap.add_argument("--check", action="store_true", help="also compare 13 steps on random data with single-step launches")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()

    if args.libs:
        child = [a for a in sys.argv[1:]]
        i = child.index("--libs")
        del child[i:i + 2]
        for rnd in range(args.rounds):
            for lib in args.libs.split(","):
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + child, env=dict(os.environ, LBM_LIB=lib),
                                   capture_output=True, text=True)
                for ln in r.stdout.splitlines():
                    print("%-24s %s" % (lib, ln), flush=True)
                if r.returncode:
                    print(lib, "FAILED", r.stderr[-2000:], flush=True)
        return
    for size in args.sizes.split(","):
        nx, ny = (int(v) for v in size.split("x"))
        for ring in (args.ring.split(",") if args.ring else [""]):
            for optset in args.opts.split(";") if args.opts else [""]:
                try:
                    res = run_case(args, nx, ny, ring, optset)
                except Exception as e:  # an option a build does not know must not end the sweep
                    print("%5dx%-5d %-5s [%s] FAILED: %s" % (nx, ny, ring or "-", optset, e), flush=True)
                    continue
                if args.json:
                    print(json.dumps(dict(res, nx=nx, ny=ny, ring=ring, opts=optset)), flush=True)
                else:
                    e = res["effective"]
                    extra = "" if "identical" not in res else "  identical=%s av_rel=%.1e" % (res["identical"], res["av_rel"])
                    print("%5dx%-5d %-5s [%-28s] %8.2f us/step %9.0f MLUPS  (fuse %d multistep %d pair %d, %d steps)%s" % (
                        nx, ny, ring or "-", optset, res["us_per_step"], res["mlups"], e["fuse"], e["multistep"], e["pair"],
                        res["steps"], extra), flush=True)


if __name__ == "__main__":
    main()

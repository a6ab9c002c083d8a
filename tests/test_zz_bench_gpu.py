"""bench.py on the GPU box, as subprocesses: the JSON contract, the self-launch of the ranks, the rehearsal of the multi-rank path
with 2 and 3 ranks on the one GPU, the one-process form, the single-rank RCCL path.  Kept in a file that sorts LAST: these
tests start several processes that share the GPU and depend on the box more than the kernel tests do; with `pytest -x` a hiccup here
must not hide the parity tests."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench_failure(r):
    """what a failed bench.py run said, without the per-process noise lines (for assertion messages)"""
    noise = ("amdgpu.ids", "socket.cpp", "[Gloo]", "OMP_NUM_THREADS", "*****")
    err = [ln for ln in r.stderr.splitlines() if not any(n in ln for n in noise)]
    return "exit code %d\nstderr:\n%s\nstdout tail:\n%s" % (r.returncode, "\n".join(err[-60:]), r.stdout[-600:])


def _records(r, rank_mode):
    """the record lines of a run: ONE on a single GPU; with ranks the headline record (`partial`) and then the full one"""
    import json
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not rank_mode:
        assert len(lines) == 1
        return lines[0]
    assert len(lines) == 2 and "partial" in lines[0] and "partial" not in lines[1]
    head, full = lines
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "result_ok", "transport_check"):
        assert key in head and head[key] == full[key], key     # the early record is the full record minus the side legs
    assert "also" not in head and "weak" not in head
    return full


def _checked(tc, transports):
    """transport_check of a clean run: every transport through all three legs"""
    assert tc["fault_injected"] is None and set(tc["legs"]) == {"small", "deep", "digest"}
    for t in transports:
        assert tc["transports"][t]["ok"] and tc["transports"][t]["failed_legs"] == []
        small, deep, dig = (tc["legs"][k]["transports"][t] for k in ("small", "deep", "digest"))
        assert small["ok"] and small["cells_max_rel"] < 2e-5 and small["av_vels_max_rel"] < 1e-4 and small["exchanges"] >= 50
        assert deep["ok"] and deep["cells_max_rel"] < 2e-5 and deep["av_vels_max_rel"] < 1e-4
        assert deep["halo_depth"] == 8 and deep["exchanges"] == 5 and deep["kernel"].startswith("d2q9_deep")
        assert dig["ok"] and len(dig["digest_rank0"]) == 32
    # the deep leg ran the kernels the 8192x8192 leg of a multi-GPU run runs: chunk pairs with the in-kernel push, ONE launch per
    # launch set — into the ring neighbours over peer stores, into the staging blocks the edge stream sends over RCCL (round 4)
    for t in ("peer", "rccl"):
        if t in transports:
            assert tc["legs"]["deep"]["transports"][t]["kernel"] == "d2q9_deep_twin x8, compact launch sets"


def test_bench_json_contract():
    """bench.py prints ONE JSON line with the driver's keys plus `roofline` and `cpu_baseline` (small grid here)"""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "8", "--nx", "1024",
                        "--ny", "1024", "--no-extra"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, _bench_failure(r)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 40 and j["warmup"] == 8 and j["higher_is_better"] is True
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and j["vs_baseline"] is None and "workload" in j["config"]
    assert abs(j["value"] - 1024 * 1024 / (j["ms_per_step"] * 1e-3) / 1e6) / j["value"] < 0.01
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf and rf["frac"] <= 1.0
    # every field recomputable from the others: model bytes (72 + 1 per cell) over the launch time
    assert rf["model_bytes_per_launch"] == 73.0 * 1024 * 1024
    assert abs(rf["achieved"] - rf["model_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) / rf["achieved"] < 0.01
    alg = rf["algorithmic"]
    assert abs(alg["gbps"] - 72.0 * 1024 * 1024 * rf["steps_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) / alg["gbps"] < 0.01
    # the issue-rate roofline of the deep window kernel: measured packed-FMA rate, lane-instructions per update from the profile
    va = rf["valu"]
    assert 5.0 < va["issue_rate_measured"] <= va["theoretical"] * 1.1 and (va["frac"] is None or 0.0 < va["frac"] < 1.0)
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 1 and cb["unit"] == "MLUPS" and cb["sample"]
    # BASELINE config 1 (128x128 on the serial CPU path, full length, through the checker) and the 1024x1024 rates
    assert cb["input_128x128_full_run"]["check_py"] == "passed" and cb["input_128x128_full_run"]["steps"] == 40000
    assert cb["input_1024x1024_rate"]["f32"]["steps"] >= 200 and cb["input_1024x1024_rate"]["f64"]["value"] > 1
    assert j["result_ok"] is True
    # result_ok has teeth: av_vels of the timed context against the oracle's record of the same steps (cpu_baseline leg)
    rc = j["result_check"]
    assert rc["compared_steps"] >= 3 and rc["av_vels_max_rel_vs_oracle"] < rc["tolerance"] == 1e-4
    # the warm-up that is not in --warmup is in the record, and so is the figure without it
    pw = j["pre_warmup"]
    assert pw["copy_launches"] == 10 and pw["valu_calib_launches"] == 40 and pw["valu_calib_ms"] > 10
    assert 0.3 * j["value"] < j["value_cold"] < 1.5 * j["value"]
    assert j["library"].startswith("lbm-hip") and "src " in j["library"] and j["launcher"].startswith("none")


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher around it: bench.py starts torch.distributed.run itself as a child
    process (the parent never touches the GPU) and relays rank 0's line.  With --gpus 2 on this one-GPU box the CHILD
    ranks must refuse with a clear message and a non-zero exit code; with --launcher torchrun and one rank the whole
    rank path (RCCL communicator, both transports checked against the oracle, strong + 1024x1024 + weak legs) runs."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LBM_BENCH_RANK_MODE", "LBM_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "2 GPUs needed, 1 visible" in r.stderr, r.stderr[-2000:]
    assert "needs a torch.distributed launch" not in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launcher", "torchrun", "--steps", "60",
                        "--warmup", "12", "--nx", "2048", "--ny", "1024", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, _bench_failure(r)
    j = _records(r, True)
    assert j["n_gpus"] == 1 and j["result_ok"] is True and j["launcher"].startswith("torch.distributed.run started by bench.py")
    assert set(j["transports"]) == {"peer", "rccl"} and j["rccl_world_size"] == 1
    _checked(j["transport_check"], ["peer", "rccl"])
    assert j["transport_check"]["legs"]["digest"]["compared_with"] == "each other"     # peer and RCCL, bit for bit
    assert j["also"]["value"] > 1000 and j["weak"]["scaling"] == "weak" and j["weak"]["value"] > 1000
    assert j["weak"]["per_gpu"] == j["weak"]["value"] and len(j["weak"]["per_rank_launch_set_us"]) == 1


@pytest.mark.parametrize("nranks", [2, 3])
def test_bench_rehearsal_of_the_multi_rank_path(nranks):
    """every N > 1 code path of bench.py on the one GPU of this box: `python bench.py --gpus N` (LBM_BENCH_REHEARSAL=1) starts N
    ranks itself; the ranks are real processes that own row slabs of ONE grid, rendezvous over gloo, map their ring
    neighbours through HIP IPC and move halo rows by peer stores (RCCL refuses two ranks per device, so the rehearsal
    runs without communicator and adds the velocity records up itself).  Checked: the transport against the oracle
    (transport_check), the timed record against the oracle (result_check), the per-rank launch-set gather, the
    1024x1024 strong leg and the weak leg — with 3 ranks also uneven slabs (1024 = 342 + 341 + 341 rows)."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LBM_BENCH_RANK_MODE", "LBM_BENCH_CHILD")}
    env["LBM_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "24", "--warmup", "8",
                        "--nx", "2048", "--ny", "1024", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, _bench_failure(r)
    j = _records(r, True)
    assert j["n_gpus"] == nranks and j["result_ok"] is True and "NOT a benchmark" in j["rehearsal"]
    assert j["config"]["partition"] == "rows x%d" % nranks and abs(j["config"]["rows_per_gpu"] - 1024 / nranks) < 1
    assert set(j["transports"]) == {"peer"} and j["transport"] == "peer" and j["rccl_world_size"] == 0
    _checked(j["transport_check"], ["peer"])
    assert j["transport_check"]["legs"]["digest"]["compared_with"] == "undivided grid"
    assert "8192x%d" % (768 * nranks) in j["transport_check"]["legs"]["deep"]["workload"]
    assert j["result_check"]["compared_steps"] >= 3 and j["result_check"]["av_vels_max_rel_vs_oracle"] < 1e-4
    pr = j["per_rank_launch_set_us"]
    assert [p["rank"] for p in pr] == list(range(nranks)) and all(p["transport"] == "peer" and p["sets"] >= 1 for p in pr)
    assert sum(p["rows"] for p in pr) == 1024
    assert j["also"]["value"] > 100 and len(j["also"]["per_rank_launch_set_us"]) == nranks
    assert j["weak"]["scaling"] == "weak" and j["weak"]["workload"].startswith("2048x%d" % (1024 * nranks)) and j["weak"]["value"] > 100
    assert j["value_cold"] > 100


@pytest.mark.parametrize("legs", ["deep", "small,digest"])
def test_bench_transport_check_catches_a_stale_halo(legs):
    """the checks have teeth: with ONE halo exchange of the named check legs delivering nothing (library test hook
    debug_stale_exchange: the flag words go up, the rows stay what they were — what a transport that loses halo rows over xGMI
    would look like) the transport fails exactly those legs, is neither timed nor reported, and the run ends with exit code 1
    and an error record"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LBM_BENCH_RANK_MODE", "LBM_BENCH_CHILD")}
    env["LBM_BENCH_REHEARSAL"] = "1"
    env["LBM_BENCH_FAULT"] = "stale_halo@" + legs
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8",
                        "--nx", "2048", "--ny", "1024", "--no-cpu-baseline", "--no-cold"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, _bench_failure(r)
    j = lines[0]
    assert j["value"] is None and j["result_ok"] is False and "no halo transport passed" in j["error"]
    tc = j["transport_check"]
    assert tc["fault_injected"] == sorted(legs.split(",")) and tc["transports"]["peer"]["ok"] is False
    assert sorted(tc["transports"]["peer"]["failed_legs"]) == sorted(legs.split(","))
    for leg in ("small", "deep"):
        t = tc["legs"][leg]["transports"]["peer"]
        assert t["ok"] == (leg not in legs) and (t["cells_max_rel"] > 1e-3) == (leg in legs)    # stale rows: errors of the noise's size
    d = tc["legs"]["digest"]["transports"]["peer"]
    assert d["ok"] == ("digest" not in legs) and (("ranks_that_differ" in d) == ("digest" in legs))


def test_bench_falls_back_from_staged_to_two_stream_rccl_sets():
    """RCCL's launch sets are staged since round 4 (one launch per set, edge units push into staging blocks, the exchange behind a
    stream wait-value).  If that form fails a check on the machine at hand, bench.py puts the two-stream sets of round 3 ("rccl2":
    option compact 0) through the same three checks and times those instead.  Rehearsed on the RCCL ring of one with a lost
    exchange injected into the staged contexts only"""
    import json
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LBM_BENCH_RANK_MODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", LBM_BENCH_FAULT="stale_halo@deep:rccl")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "24", "--warmup", "8", "--nx", "8192", "--ny", "1024", "--transport", "rccl",
                        "--no-cpu-baseline", "--no-extra", "--no-cold"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, _bench_failure(r)
    j = _records(r, True)
    tc = j["transport_check"]
    assert tc["transports"]["rccl"]["ok"] is False and tc["transports"]["rccl"]["failed_legs"] == ["deep"]
    assert tc["transports"]["rccl2"]["ok"] is True and tc["legs"]["deep"]["transports"]["rccl2"]["kernel"] == "d2q9_deep x8, two streams"
    assert set(j["transports"]) == {"rccl2"} and j["transport"] == "rccl2" and j["value"] > 1000 and j["result_ok"] is True


def test_bench_one_process_form_rehearsal():
    """--launcher one-process: ONE process drives N row slabs (lbm_create(ndev = N)) — the form
    bench.py falls back to where torch.distributed.run is missing; rehearsed with 4 slabs on the one GPU"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LBM_BENCH_RANK_MODE", "LBM_BENCH_CHILD")}
    env["LBM_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launcher", "one-process", "--steps", "24",
                        "--warmup", "8", "--nx", "2048", "--ny", "1024", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, _bench_failure(r)
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 4 and j["result_ok"] is True and j["launcher"].startswith("one process, lbm_create(ndev=4)")
    assert j["config"]["rows_per_gpu"] == 256 and j["per_rank_launch_set_us"][0]["sets"] >= 1


def test_bench_one_process_per_gpu_path_single_rank():
    """the driver's multi-GPU launch line (torch.distributed.run, one rank per GPU, RCCL) with ONE rank that is its
    own ring neighbour: torch.distributed's nccl backend and the library's RCCL communicator (ncclCommInitRank from
    the broadcast id, grouped send/recv halos, all-reduce of the velocity sums) in one process on the real GPU"""
    import json
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LBM_BENCH_RANK_MODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "60", "--warmup", "12", "--nx", "2048", "--ny", "1024",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, _bench_failure(r)
    j = _records(r, True)
    assert j["n_gpus"] == 1 and j["result_ok"] is True and j["value"] > 1000
    # 2048x1024 = 2M cells with halo rows: the five-step chunk pairs, one launch per launch set — over peer stores and (staged) over RCCL
    assert j["roofline"]["steps_per_launch"] == 5
    # both halo transports were measured on the ring of one; per-rank launch-set timings explain the record
    assert set(j["transports"]) == {"peer", "rccl"} and j["transport"] in j["transports"] and j["rccl_world_size"] == 1
    pr = j["per_rank_launch_set_us"]
    assert len(pr) == 1 and pr[0]["sets"] >= 4 and pr[0]["interior_us"] > 0 and pr[0]["set_period_us"] > 0
    # (compact launch sets under both transports: ONE launch per set, reported as the interior launch; no separate edge launch)
    assert pr[0]["transport"] == j["transport"] and pr[0]["edge_us"] == 0
    # and the reference's 1024x1024 input row-partitioned over the same ranks (BASELINE config 4's leg of a multi-GPU record)
    assert j["also"]["value"] > 1000 and j["also"]["halo_depth"] >= 3 and len(j["also"]["per_rank_launch_set_us"]) == 1
    # before any timing every transport reproduced the oracle on the 1024x1024 obstacles from a random state (a transport
    # delivering stale or misplaced halo rows would not), and the weak-scaling leg of config 5 is in the same line
    _checked(j["transport_check"], ["peer", "rccl"])
    assert j["weak"]["value"] > 1000 and j["launcher"].startswith("external")

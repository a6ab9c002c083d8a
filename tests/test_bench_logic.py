"""bench.py's host logic that needs no GPU: the check `result_ok` hangs on, and the self-launch of the ranks."""
import importlib.util
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_result_check_has_teeth():
    """a record that deviates from the oracle's by more than 1e-4 — or is too short, or not finite — is refused"""
    bench = load_bench()
    ref = np.linspace(1e-6, 2e-5, 12)
    assert bench.result_check(ref * (1 + 5e-5), ref)[0] is True
    assert bench.result_check(ref * (1 + 3e-4), ref)[0] is False
    half = ref.copy()
    half[5:] *= 0.5                       # "a kernel that skipped half its levels"
    ok, detail = bench.result_check(half, ref)
    assert ok is False and detail["compared_steps"] == 12
    assert bench.result_check(ref[:1], ref)[0] is False
    bad = ref.copy()
    bad[3] = np.nan
    assert bench.result_check(bad, ref)[0] is False
    # only the steps both records hold are compared
    assert bench.result_check(np.concatenate([ref, ref[-1:] * 3]), ref)[0] is True


def test_oracle_rate_returns_the_av_record(oracle_f32):
    """the cpu_baseline leg keeps the oracle's av_vels (it used to throw them away): same values as oracle.run"""
    bench = load_bench()
    ob = bench.cavity(64, 48)
    av = []
    n, el = bench.oracle_rate("f32", 64, 48, ob, 0.005, 0.0, 5, av_out=av)
    assert n >= 2 and el > 0 and len(av) == n + 1
    p = oracle_f32.make_params(64, 48, len(av), 10, 0.1, 0.005, 1.85)
    oracle_f32.set_obstacles(p, ob)
    cells = oracle_f32.init_cells(p)
    ref = oracle_f32.run(p, cells, ob, len(av))
    assert np.allclose(av, ref, rtol=1e-6, atol=0)


def test_traffic_of_another_build_is_refused(tmp_path, monkeypatch):
    """profiles/traffic.json names the library build it was measured on; counters of another build are not attached"""
    import json
    bench = load_bench()
    root = tmp_path
    (root / "profiles").mkdir()
    ent = {"hbm_bytes_per_launch": 5.0e9, "library_version": "lbm-hip X src aaaa", "source": "somewhere",
           "evidence": {"valu_issue_share": 0.7, "valu_lane_instr_per_cell_step": 70.0}}
    (root / "profiles" / "traffic.json").write_text(json.dumps({"8192x8192/deep": ent}))
    monkeypatch.setattr(bench, "ROOT", str(root))
    rf = {"traffic": None}
    bench.attach_traffic(rf, 3.6e11, 8192, 8192, True, False, 0, 4.9e9, 1.4e-3, 37.0, "lbm-hip X src bbbb")
    assert rf["traffic"] is None and "traffic_refused" in rf and "aaaa" in rf["traffic_refused"]["reason"]
    rf = {"traffic": None}
    bench.attach_traffic(rf, 3.6e11, 8192, 8192, True, False, 0, 4.9e9, 1.4e-3, 37.0, "lbm-hip X src aaaa")
    assert rf["traffic"] == 5.0e9 and rf["limited_by"] == "valu_issue" and abs(rf["valu"]["frac"] - 70.0 * 3.6e11 / 1e12 / 37.0) < 1e-3


def test_self_launch_without_gpu_fails_in_the_child():
    """`python bench.py --gpus 2` with no launcher around it starts torch.distributed.run as a child process; on this
    CPU-only box the child ranks refuse ("needs a GPU"), the parent relays that and exits non-zero without a result line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LBM_BENCH_RANK_MODE", "LBM_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr or "GPUs needed" in r.stderr, r.stderr[-2000:]
    assert "torch.distributed.run exited with code" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]

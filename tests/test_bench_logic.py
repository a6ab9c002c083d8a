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
    rf = {"traffic": None, "steps_per_launch": 8.0}
    bench.attach_traffic(rf, 3.6e11, 8192, 8192, True, False, 0, 4.9e9, 1.4e-3, 37.0, "lbm-hip X src aaaa")
    assert rf["traffic"] == 5.0e9 and rf["limited_by"] == "valu_issue" and abs(rf["valu"]["frac"] - 70.0 * 3.6e11 / 1e12 / 37.0) < 1e-3
    assert abs(rf["traffic_frac"] - 5.0e9 / 1.4e-3 / 1e9 / 8000.0) < 1e-3      # (an entry without launch record: this run's launches)


def test_traffic_is_priced_on_the_launches_it_was_counted_on(tmp_path, monkeypatch):
    """counter bytes of the 8-step launches of a profile must not be divided by the duration of another run's 6.67-step
    launches: the entry carries its own launch duration, and without one only a run of the same depth is priced"""
    import json
    bench = load_bench()
    (tmp_path / "profiles").mkdir()
    ent = {"hbm_bytes_per_launch": 5.5e9, "library_version": "v", "source": "s", "steps_per_launch": 8, "launch_us": 1150.0,
           "evidence": {"valu_issue_share": 0.71, "valu_lane_instr_per_cell_step": 55.0}}
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps({"8192x8192/deep_twin": ent}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rf = {"traffic": None, "steps_per_launch": 6.667}
    bench.attach_traffic(rf, 4.4e11, 8192, 8192, True, True, 0, 4.9e9, 1.0e-3, 37.0, "v")
    assert abs(rf["traffic_frac"] - 5.5e9 / 1150e-6 / 1e9 / 8000.0) < 1e-3 and rf["traffic_launch"]["steps_per_launch"] == 8
    del ent["launch_us"]
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps({"8192x8192/deep_twin": ent}))
    rf = {"traffic": None, "steps_per_launch": 6.667}
    bench.attach_traffic(rf, 4.4e11, 8192, 8192, True, True, 0, 4.9e9, 1.0e-3, 37.0, "v")
    assert rf["traffic"] == 5.5e9 and rf["traffic_frac"] is None and "limited_by" not in rf


def test_seeded_state_is_a_function_of_the_global_row():
    """every rank computes the rows it needs and any two ranks the same values for the same global row; no two neighbouring rows
    and no two rows a slab height apart are equal"""
    bench = load_bench()
    whole = bench.seeded_state(64, np.arange(200))
    part = bench.seeded_state(64, np.arange(150, 190) % 200)
    assert whole.dtype == np.float32 and whole.shape == (9, 200, 64) and np.array_equal(whole[:, 150:190], part)
    assert not np.array_equal(whole[:, 0], whole[:, 1]) and not np.array_equal(whole[:, 3], whole[:, 3 + 128])
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4).reshape(9, 1, 1) * 0.1
    assert np.all(np.abs(whole / w - 1.0) <= 0.1001) and np.std(whole / w) > 0.04


def test_band_oracle_equals_the_oracle_of_the_whole_grid(oracle_f32_omp):
    """the per-rank oracle of transport_check: rows [y0, y1) computed on a band of rows + nsteps + 1 either side give exactly the
    rows, and the per-step velocity sums exactly the share, of the oracle stepped on the whole periodic grid — also for the slab
    that holds the accelerated row ny-2, for a band that wraps around the grid, and where the band is the whole grid"""
    bench = load_bench()
    nx, ny, nsteps = 48, 120, 9
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[:, 0] = ob[:, -1] = 1
    ob[40:44, 10:14] = 1
    ob[117, 20:30] = 1
    orc = oracle_f32_omp
    p = orc.make_params(nx, ny, nsteps, 10, 0.1, 0.005, 1.85)
    orc.set_obstacles(p, ob)
    cells = bench.seeded_state(nx, np.arange(ny))
    tmp = np.empty_like(cells)
    raw_rows = []
    for t in range(nsteps):      # the whole grid, keeping every step's per-slab sums
        orc.accelerate_flow(p, cells, ob)
        raw_rows.append([orc.timestep_rows(p, cells, tmp, ob, a, b) for a, b in ((0, 40), (40, 80), (80, 120))])
        cells, tmp = tmp, cells
    raw_rows = np.array(raw_rows)
    for i, (a, b) in enumerate(((0, 40), (40, 80), (80, 120))):
        got, raw = bench.band_oracle(orc, nx, ny, ob, a, b, nsteps, 0.1, 0.005, 1.85)
        assert np.array_equal(got, cells[:, a:b]) and np.array_equal(raw, raw_rows[:, i])
    got, raw = bench.band_oracle(orc, nx, ny, ob, 10, 115, nsteps, 0.1, 0.005, 1.85)    # band = the whole grid
    assert np.array_equal(got, cells[:, 10:115])


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

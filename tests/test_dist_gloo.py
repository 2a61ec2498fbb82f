"""The N > 1 path of bench.py on CPU: two ranks over gloo (127.0.0.1), grid points sharded
with no data-path collective; --dry-run skips only the GPU launches."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


N3 = 137 * 1800 * 3600
INNER = 1800 * 3600


def _run_ranks(world, *extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "3", "--warmup", "1", "--dry-run", *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # only rank 0 reports
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1
    assert d["dry_run"] is True and d["value"] is None  # a dry run never reports a throughput
    # the stamps of every rank reach rank 0: the region between the barriers, and how far apart the ranks left the first
    assert d["timed_region_ms"] > 0 and d["barrier_skew_ms"] >= 0 and d["end_skew_ms"] >= 0
    assert d["barrier_skew_ms"] < 5000, d  # gloo ranks of one host leave a barrier within seconds of one another
    return d


@pytest.mark.parametrize("world", [2, 4, 8])
def test_default_for_n_gt_1_is_config_5_strong_scaling(world):
    """BASELINE.json config 5: ONE 3600x1800x137 field sharded by grid point across the GPUs."""
    d = _run_ranks(world)
    assert d["scaling"] == "strong"
    cfg = d["config"]
    assert cfg["points_total"] == N3 and cfg["points_per_gpu"] == N3 // world
    ranks = cfg["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(world))
    assert sum(r["points"] for r in ranks) == N3  # the shards cover the field exactly once


def test_two_rank_gloo_dry_run_weak():
    d = _run_ranks(2, "--scaling", "weak")
    assert d["scaling"] == "weak"
    assert d["config"]["points_total"] == 2 * N3 and [r["points"] for r in d["config"]["per_rank"]] == [N3, N3]


@pytest.mark.parametrize("pmode", ["level", "hybrid"])
def test_strong_scaling_with_level_pressure_cuts_on_level_boundaries(pmode):
    d = _run_ranks(8, "--pmode", pmode)
    assert d["scaling"] == "strong" and d["config"]["shard_cut"] == "levels"
    pts = [r["points"] for r in d["config"]["per_rank"]]
    assert sum(pts) == N3 and all(n % INNER == 0 for n in pts)
    assert sorted({n // INNER for n in pts}) == [17, 18]  # 137 levels over 8 GPUs


def test_plan_shard_covers_the_field_for_every_mode():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
    import bench

    for world in (1, 2, 3, 4, 8):
        for workload, pmode in (("full", "field"), ("full", "level"), ("p3", "hybrid"), ("geopotential", "hybrid"),
                                ("hybrid_levels", "field")):
            sh = [bench.plan_shard(workload, pmode, "strong", r, world) for r in range(world)]
            assert sum(s["n_local"] for s in sh) == N3 and all(s["n_total"] == N3 for s in sh)
            if workload in bench.COLUMN_WORKLOADS:
                assert [s["col0"] for s in sh[1:]] == [s["col1"] for s in sh[:-1]] and sh[-1]["col1"] == INNER
                assert all(s["n_local"] == 137 * (s["col1"] - s["col0"]) for s in sh)
            elif pmode != "field":
                assert [s["lev0"] for s in sh[1:]] == [s["lev1"] for s in sh[:-1]] and sh[-1]["lev1"] == 137
                assert all(s["first"] == s["lev0"] * INNER for s in sh)
            else:
                assert all(s["first"] % 16 == 0 for s in sh)  # 64-B aligned shard starts keep the float4 path
                assert [s["first"] + s["n_local"] for s in sh[:-1]] == [s["first"] for s in sh[1:]]
            weak = bench.plan_shard(workload, pmode, "weak", 0, world)
            assert weak["n_local"] == N3 and weak["n_total"] == N3 * world


def test_shards_of_a_field_cover_it_exactly():
    sys.path.insert(0, os.path.join(ROOT, "earthkit-meteo_amd"))
    from ekm_hip.device import shard_bounds

    n3 = 137 * 1800 * 3600
    for world in (1, 2, 4, 8):
        b = shard_bounds(n3, world)
        assert sum(hi - lo for lo, hi in b) == n3
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 16 * world


def test_bench_json_contract_single_rank():
    """Every key the driver's bench contract names is present (dry run: no GPU, value null)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    meta = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == meta["metric"]

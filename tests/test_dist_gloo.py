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


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_gloo_dry_run(scaling):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "1", "--dry-run", "--scaling", scaling]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # only rank 0 reports
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["dry_run"] is True and d["value"] is None  # a dry run never reports a throughput
    n3 = 137 * 1800 * 3600
    assert d["config"]["points_per_gpu"] == (n3 if scaling == "weak" else n3 // 2)


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

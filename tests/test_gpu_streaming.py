"""Host-side data movement on a real MI355X: the streamed NumPy path (bounded device working set, prefault,
lanes), cross-stream ordering, the block cache, and DLPack exchange with a FOREIGN ROCm library (torch is
used here -- in the tests only -- as that foreign producer / consumer; the product never imports it)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
np.seterr(all="ignore")


def _fields(nlev, npts, dtype=np.float32, seed=5):
    from oracle import synthetic

    t, q, p, _ = synthetic.make_fields(nlev, npts, dtype=dtype, seed=seed)
    return t, q, p


def _device_path(ek, func, arrays, **kw):
    d = [ek.to_device(a) for a in arrays]
    out = getattr(ek.thermo, func)(*d, **kw)
    out = out if isinstance(out, tuple) else (out,)
    return [o.to_host() for o in out]


def test_large_multi_output_numpy_call_is_bit_equal_to_the_device_path(ek):
    """>= 256 MB of NumPy input takes the streamed path (slices on lane streams, result pages prefaulted by a
    helper thread while data is in flight).  The prefault must never alter data that has already landed:
    every one of the six outputs must equal the DeviceArray path bit for bit."""
    t, q, p = _fields(16, 1 << 21)  # 3 x 128 MiB in, 6 x 128 MiB out
    assert t.nbytes * 3 >= 256 << 20
    want = _device_path(ek, "pipeline_full", (t, q, p))
    for rep in range(3):
        got = ek.thermo.pipeline_full(t, q, p)
        for k, (g, w) in enumerate(zip(got, want)):
            assert g.dtype == w.dtype and g.shape == w.shape
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), f"rep {rep} output {k} differs"


def test_kernels_writing_straight_into_pinned_results_give_the_same_bits(ek, monkeypatch):
    """EKM_DIRECT_RESULTS=1 (round 6, opt-in): the output pointers of a launch are the pooled pinned host blocks, nothing is
    downloaded.  The streamed call (>= 256 MB in), a single-launch call (the pool is used from 32 MiB of results) and a ragged
    size through both -- the bits of the DeviceArray path."""
    from ekm_hip import _engine

    monkeypatch.setattr(_engine, "_DIRECT_OUT", True)
    for nlev, npts, func in ((16, 1 << 21, "pipeline_full"), (16, (1 << 21) + 1028, "pipeline_svp_td_rh"), (3, (1 << 21) + 3, "pipeline_full")):
        t, q, p = _fields(nlev, npts)
        want = _device_path(ek, func, (t, q, p))
        for rep in range(2):
            got = getattr(ek.thermo, func)(t, q, p)
            for k, (g, w) in enumerate(zip(got, want)):
                assert g.dtype == w.dtype and g.shape == w.shape
                assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), f"{func} {nlev}x{npts} rep {rep} output {k} differs"
    # a level vector for the pressure and float64 inputs under a float32 override: still the same bits
    t, q, p = _fields(16, 1 << 21)
    pl = p[:, :1].copy()
    want = _device_path(ek, "wet_bulb_temperature_from_specific_humidity", (t, q, pl), ept_method="ifs", t_method="bisect")
    got = ek.thermo.wet_bulb_temperature_from_specific_humidity(t, q, pl, ept_method="ifs", t_method="bisect")
    assert np.array_equal(got.view(np.uint32), want[0].view(np.uint32))


def test_converted_operands_share_the_slice_pipeline(ek):
    """float64 inputs with a float32 override are converted copies: they travel with the other operands of a slice, rows
    of 8 MiB + 4112 B put the slice boundaries off the page grid, and the results are the bits of the device path."""
    from ekm_hip import _engine, _streamed  # noqa: F401

    t, q, p = _fields(16, (1 << 21) + 1028)
    want = _device_path(ek, "pipeline_svp_td_rh", (t, q, p))
    for rep in range(2):
        got = ek.thermo.pipeline_svp_td_rh(t, q, p)
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), f"rep {rep} output {k} differs"
    got64 = _streamed._run_streamed("potential_temperature", (t.astype(np.float64), p), (), None, np.float32, [ek.current_device()])[0]
    assert np.array_equal(got64, _device_path(ek, "potential_temperature", (t, p))[0])


def test_big_numpy_results_live_in_pooled_pinned_memory(ek):
    """Results of a streamed NumPy call come in pinned host memory from a recycling pool (device.pinned_empty): they are
    ordinary writable NumPy arrays, results of successive calls never alias while both are alive, a block returns to
    the pool only when the array AND its views are gone, and inputs in pinned memory work the same."""
    import gc

    from ekm_hip import device

    t, q, p = _fields(16, 1 << 21)
    want = _device_path(ek, "pipeline_svp_td_rh", (t, q, p))
    first = ek.thermo.pipeline_svp_td_rh(t, q, p)
    second = ek.thermo.pipeline_svp_td_rh(t, q * np.float32(0.5), p)   # while `first` is alive: other blocks
    addr = lambda a: a.__array_interface__["data"][0]  # noqa: E731
    assert len({addr(a) for a in first + second}) == 6 and all(a.flags.writeable and a.flags.c_contiguous for a in first)
    for g, w in zip(first, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    view = first[0][3:5]                                   # a view keeps the whole block alive
    kept = addr(first[0])
    held = device._pinned.handed_out
    del first, g, w
    gc.collect()
    assert device._pinned.handed_out == held - 2 * device._pinned.bucket(t.nbytes)   # two of the three came back
    third = ek.thermo.pipeline_svp_td_rh(t, q, p)
    assert kept not in {addr(a) for a in third}           # ... the viewed block was not handed out again
    assert np.array_equal(view, want[0][3:5])
    third[0][0, :8] = 0.0                                  # writable like any array
    # inputs in pinned memory (ekm_hip.pinned_empty): same results
    pin = [ek.pinned_empty(a.shape, a.dtype) for a in (t, q, p)]
    for dst, src in zip(pin, (t, q, p)):
        dst[...] = src
    fourth = ek.thermo.pipeline_svp_td_rh(*pin)
    for g, w in zip(fourth, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    del second, third, fourth, view, pin, g, w, dst, src
    gc.collect()
    ek.empty_cache()
    assert device._pinned.cached == 0 and device._pinned.handed_out == 0


def test_streaming_stays_within_a_capped_device_budget(ek, monkeypatch):
    """fields whose working set exceeds the device budget stream through in slices, two or
    more in flight, device blocks recycled between slices.  Budget capped at 96 MiB against a 768 MiB
    working set; the peak of live device bytes must stay under the cap and the result must be unchanged."""
    from ekm_hip import _engine, _streamed  # noqa: F401

    t, q, p = _fields(64, 1 << 19)  # 3 x 128 MiB in, 3 x 128 MiB out = 768 MiB working set
    want = _device_path(ek, "pipeline_svp_td_rh", (t, q, p))
    cap = 96 << 20
    monkeypatch.setenv("EKM_STREAM_BUDGET_BYTES", str(cap))
    ek.empty_cache()
    ek.memory_stats(reset_peak=True)
    base = ek.memory_stats()["live_bytes"]
    lanes, nslices = _streamed.plan_slices(64, 6 * (1 << 19) * 4, cap, overhead=6 * _streamed._BLOCK_OVERHEAD)
    assert lanes >= 2 and nslices > lanes  # more slices than lanes: blocks are recycled
    got = ek.thermo.pipeline_svp_td_rh(t, q, p)
    st = ek.memory_stats()
    peak, foot = st["peak_live_bytes"] - base, st["peak_footprint_bytes"] - base
    print(f"streaming: lanes={lanes} slices={nslices} peak live {peak / 2**20:.1f} MiB, peak live + cached "
          f"{foot / 2**20:.1f} MiB of cap {cap / 2**20:.0f} MiB")
    assert 0 < peak <= cap, (peak, cap)
    # live + CACHED: slices differ by one row, but every slice reserves the longest slice's block size, so a lane
    # takes back exactly the blocks it released and nothing piles up in the cache
    assert foot <= cap, (foot, cap)
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    # a budget that cannot hold two single-row slices is an error, not an OOM
    monkeypatch.setenv("EKM_STREAM_BUDGET_BYTES", str(1 << 20))
    with pytest.raises(ek.EkmError, match="streaming budget"):
        ek.thermo.pipeline_svp_td_rh(t, q, p)


def test_level_vector_operand_is_sliced_with_the_fields(ek):
    t, q, p = _fields(32, 1 << 20)
    plev = np.ascontiguousarray(p[:, :1])  # (32, 1): one pressure per level
    pfull = np.ascontiguousarray(np.broadcast_to(plev, t.shape))
    want = _device_path(ek, "potential_temperature", (t, pfull))[0]
    got = ek.thermo.potential_temperature(t, plev)  # 2 x 128 MiB... t alone is 128 MiB: force the streamed path
    from ekm_hip import _engine, _streamed  # noqa: F401

    got2 = _streamed._run_streamed("potential_temperature", (t, plev), (), None, None, [ek.current_device()])[0]
    assert np.array_equal(got, want) and np.array_equal(got2, want)


def test_float16_and_dtype_override_do_not_depend_on_size(ek):
    """the result dtype of float16 input must not depend on which path the size selects."""
    small = np.linspace(250, 300, 1000).astype(np.float16)
    big = np.resize(small, 6 << 20)  # 12 MiB fp16 -> 24 MiB fp32 result: the pretouch path
    for a in (small, big):
        out = ek.thermo.saturation_vapour_pressure(a)
        assert out.dtype == np.float16, (a.size, out.dtype)
    from ekm_hip import _engine, _streamed  # noqa: F401

    t, q, p = _fields(8, 1 << 16)
    o = _streamed._run_streamed("potential_temperature", (t, p), (), None, np.float64, [ek.current_device()])[0]
    assert o.dtype == np.float64


def test_cross_stream_consumer_waits_for_the_producer(ek):
    """An array produced on one non-blocking stream and consumed on another is ordered on the device
    (DeviceArray.on -> order_streams); results must equal the single-stream chain."""
    t, q, p = _fields(8, 1 << 20)
    want = _device_path(ek, "potential_temperature", (_device_path(ek, "dewpoint_from_specific_humidity", (q, p))[0], p))[0]
    s1, s2 = ek.stream_create(), ek.stream_create()
    try:
        for rep in range(5):
            ek.set_stream(s1)
            dq, dp = ek.to_device(q), ek.to_device(p)
            td = ek.thermo.dewpoint_from_specific_humidity(dq, dp)       # produced on s1 (asynchronous)
            ek.set_stream(s2)
            th = ek.thermo.potential_temperature(td, dp)                 # consumed on s2: must wait for s1
            got = th.to_host()
            assert np.array_equal(got, want), f"rep {rep}"
            del td, th, dq, dp   # blocks go back under the stream they were last used on
    finally:
        ek.set_stream(None)
        ek.synchronize()
        ek.stream_destroy(s1)
        ek.stream_destroy(s2)


def test_arrays_outlive_the_stream_they_were_computed_on(ek):
    """"compute on a temporary stream, destroy it, keep the results".  stream_destroy re-files
    the live arrays of that stream under the default stream (their work is complete after its sync), so a later
    use on another stream never records an event on the destroyed handle."""
    t, q, p = _fields(4, 1 << 18)
    want = _device_path(ek, "potential_temperature", (t, p))[0]
    s1 = ek.stream_create()
    ek.set_stream(s1)
    dt, dp = ek.to_device(t), ek.to_device(p)
    th = ek.thermo.potential_temperature(dt, dp)          # result (and inputs) last used on s1
    ek.stream_destroy(s1)                                  # also resets this thread's current stream
    assert ek.current_stream() is None and th._alloc.stream is None and dt._alloc.stream is None
    assert np.array_equal(th.to_host(), want)              # default stream
    s2 = ek.stream_create()
    try:
        ek.set_stream(s2)
        again = ek.thermo.potential_temperature(dt, dp)   # the kept inputs on a third stream
        assert np.array_equal(again.to_host(), want)
        assert np.array_equal(th.to_host(), want)
    finally:
        ek.set_stream(None)
        ek.stream_destroy(s2)


def test_scalar_pressure_with_ragged_length_takes_the_vector_path(ek):
    """a scalar operand with n % 4 != 0 runs the aligned kernel (its ragged tail element-wise);
    and a level vector too long for the LDS left beside the bisection table is materialised, not a launch error."""
    rng = np.random.default_rng(5)
    for n in (1 << 16) + np.array([1, 2, 3]):
        t = (250 + 50 * rng.random(int(n))).astype(np.float32)
        got = ek.thermo.potential_temperature(t, 85000.0)
        want = ek.thermo.potential_temperature(t, np.full(int(n), 85000.0, np.float32))
        assert np.array_equal(got, want)
    nlev = 12000  # 48 KB of fp32 levels: more than the 32 KiB the host layer lets level vectors use
    t = (250 + 50 * rng.random((nlev, 8))).astype(np.float32)
    q = np.full_like(t, 0.004)
    plev = np.linspace(30000, 100000, nlev, dtype=np.float32).reshape(nlev, 1)
    got = ek.thermo.wet_bulb_temperature_from_specific_humidity(t, q, plev)  # bisection (LDS table) + level vector
    want = ek.thermo.wet_bulb_temperature_from_specific_humidity(t, q, np.broadcast_to(plev, t.shape).copy())
    assert np.array_equal(got, want, equal_nan=True)


def test_lane_streams_are_bounded_and_releasable(ek):
    from ekm_hip import _engine, _streamed  # noqa: F401

    t, q, p = _fields(24, 1 << 18)
    for rows in (24, 17, 9):  # different leading-axis lengths must not create new streams per length
        _streamed._run_streamed("potential_temperature", (t[:rows], p[:rows]), (), None, None, [ek.current_device()])
    assert 0 < len(_streamed._streams) <= _streamed._MAX_LANES + 1  # the lanes and the one download stream
    ek.release_streams()
    assert len(_streamed._streams) == 0
    out = ek.thermo.potential_temperature(t, p)  # still works afterwards (streams are re-created on demand)
    assert np.isfinite(out).all()


# ---- DLPack with a foreign ROCm library -----------------------------------------------------------------
def test_dlpack_with_torch_as_foreign_producer_and_consumer(ek):
    """torch tensor -> ek.from_dlpack (zero copy) -> HIP kernel -> torch.from_dlpack(result) (zero copy), with the
    stream hand-over of the array-API protocol in both directions: the route the reference's
    `array_namespace(*inputs)` dispatch (thermo/array/thermo.py:826) offers to Torch users.  Runs in a child
    process (tests/_dlpack_torch_child.py) that imports torch BEFORE libekm_thermo.so is loaded: both link
    libamdhip64.so.7, and the process must use one HIP runtime for the two libraries to share pointers and
    streams (torch loaded second finds no device in this image)."""
    import subprocess
    import sys

    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dlpack_torch_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:])
    if r.returncode == 77:
        pytest.skip(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "torch unavailable")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DLPACK_TORCH_OK" in r.stdout


def _run_two_ranks(ek, extra, shared=True, levels="16"):
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--levels", levels, "--sustain", "0.2"] + extra
    if shared and ek.device_count() < 2:
        cmd.append("--allow-shared-device")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    return json.loads(line[0])


def test_bench_more_ranks_than_devices_reports_no_value(ek):
    """More ranks than GPUs without --allow-shared-device: the line says so and carries no value."""
    if ek.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    d = _run_two_ranks(ek, ["--pmode", "level"], shared=False, levels="4")
    assert d["oversubscribed"] is True and d["value"] is None and d["hip_device_count"] == 1 and d["devices_used"] == [0]
    assert d["value_from_kernel_ms"] is None


@pytest.mark.parametrize("extra,cut", [
    ([], "grid points"),                                  # the DEFAULT an 8-GPU node runs first: p a field, flat grid-point cut
    (["--pmode", "level"], "levels"),
    (["--pmode", "hybrid"], "levels"),
    (["--workload", "geopotential"], "columns"),
], ids=["field", "level", "hybrid", "geopotential-columns"])
def test_bench_two_ranks_strong_scaling_on_one_device(ek, extra, cut):
    """bench.py under torch.distributed.run with 2 ranks sharing this GPU: the N > 1 path on real hardware in every cut it
    has -- ONE field split by grid point (config 5's default), on level boundaries (level vector / hybrid levels), by
    columns (the column workloads) -- gloo barrier only, parity of rank 0's shard AND of the last rank's shard (whose first
    point is not point 0), and the last rank's inputs equal to what the whole field holds at those points.  (Two ranks on
    one device share its bandwidth: the value is not a scaling measurement.)"""
    import argparse
    import sys

    d = _run_two_ranks(ek, extra)
    nlev, inner = 16, 1800 * 3600
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["value_from_kernel_ms"] > 0
    # the timed region and how much of it is the barrier's exit skew: value = points x steps / region
    assert d["timed_region_ms"] > 0 and 0 <= d["barrier_skew_ms"] < d["timed_region_ms"] and d["end_skew_ms"] >= 0
    assert abs(d["value"] - d["config"]["points_total"] * d["steps"] / (d["timed_region_ms"] * 1e-3)) <= 1e-3 * d["value"]
    assert d["config"]["shard_cut"] == cut
    assert d["oversubscribed"] == (ek.device_count() < 2) and d["hip_device_count"] == ek.device_count()
    assert d["parity"]["ok"] and d["parity"]["nan_mismatch"] == 0
    assert d["parity_last_rank"]["ok"] and d["parity_last_rank"]["rank"] == 1 and d["parity_last_rank"]["nan_mismatch"] == 0
    ranks = d["config"]["per_rank"]
    assert sum(x["points"] for x in ranks) == nlev * inner and all(x["kernel_ms"] > 0 for x in ranks)
    assert all(x["hip_device_count"] >= 1 for x in ranks)
    assert d["sustained"]["launches"] >= 5 and d["sustained"]["kernel_ms"] > 0
    w = d["shard_window_last_rank"]
    assert w["rank"] == 1
    if cut == "columns":
        assert w["global_col0"] >= inner // 2
        return
    # the whole field, generated here as ONE shard, must hold the last rank's values at the last rank's place
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    pmode = extra[1] if extra else "field"
    args = argparse.Namespace(workload="full", pmode=pmode, dtype="f32", scaling="strong", levels=nlev)
    sh = bench.plan_shard("full", pmode, "strong", 0, 1, nlev)
    t, q, p, plev, hyb = bench.build_inputs(args, sh, 0, nlev, np.float32, 20260313)
    g = w["global_index"]
    assert g >= nlev * inner // 2 - inner, g  # rank 1's shard starts in the second half of the field
    assert t.flat_slice(g, g + 8).to_host().tolist() == w["t"]
    assert q.flat_slice(g, g + 8).to_host().tolist() == w["q"]
    if pmode == "field":
        pp = p.flat_slice(g, g + 8)
    elif pmode == "level":
        pp = plev.to_host()[g // inner]
    else:
        pp = ek.HybridPressure(hyb["Ah"].astype(np.float32), hyb["Bh"].astype(np.float32), hyb["sph"])
    if pmode == "hybrid":  # the pipeline's theta on the whole level that holds the window
        lev, col = g // inner, g % inner
        th = ek.thermo.pipeline_full(t.to_host().reshape(nlev, inner), q.to_host().reshape(nlev, inner), pp)[0][lev, col:col + 8]
    else:
        th = ek.thermo.pipeline_full(t.flat_slice(g, g + 8), q.flat_slice(g, g + 8), pp)[0]
        th = th.to_host() if hasattr(th, "to_host") else np.asarray(th)
    assert np.asarray(th, np.float32).tolist() == w["out0"]


def test_bench_measures_its_hbm_traffic_in_the_run(ek):
    """bench.py's roofline.traffic comes from two rocprofv3 --pmc child passes of the same command made after the timed
    region (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE): on 16 levels of the field it must equal the algorithmic 36 B per
    point within 1 %.  Where the profiler is not available the line says so and falls back to the committed file."""
    import json
    import shutil
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--levels", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    roof = d["roofline"]
    assert d["parity"]["ok"] and roof["bound"] == "hbm" and d["oversubscribed"] is False
    # the streaming reference timed on the launch's own arrays: nine streams, no arithmetic; the pipeline is within a few
    # per cent of it (generous bounds: 16 levels are a short launch)
    c = roof["stream_ceiling"]
    assert c["kernel_ms"] > 0 and 0.3 < c["frac"] < 1.0 and 0.8 < c["kernel_ms_over_ceiling_ms"] < 1.5, c
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH: " + roof["traffic_source"][:80])
    assert roof["traffic_source"].startswith("measured in this run"), roof["traffic_source"]
    algorithmic = roof["bytes_per_point"] * roof["points_per_launch"]
    print(f"traffic {roof['traffic']:.4g} B vs algorithmic {algorithmic:.4g} B")
    assert abs(roof["traffic"] / algorithmic - 1.0) < 0.01
    # the VALU side comes from a third child pass (SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32): executed instructions per point
    assert roof["valu_source"].startswith("measured in this run"), roof["valu_source"]
    c = roof["valu_counters"]
    assert 100 < c["SQ_INSTS_VALU"] < 300 and 10 < c["SQ_INSTS_VALU_TRANS_F32"] < 40, c
    assert 0.2 < roof["valu_frac"] < roof["frac"] < 1.0  # the six-output pipeline is HBM-bound: the HBM fraction is the larger
    assert d["sustained"]["launches"] >= 3 and d["sustained"]["seconds"] >= 1.9 and 0.3 < d["sustained"]["frac"] < 1.0
    # round 6: parity on EVERY point of the 8-level slab of the timed arrays with the wet-bulb's regime census,
    # the kernel on three independently allocated buffer sets, and the PCIe-inclusive figure beside `value`
    p = d["parity"]
    assert p["points"] == 8 * 1800 * 3600 and p["over"] == 0 and p["nan_mismatch"] == 0 and p["excluded_points"] == 0, p
    assert p["regime_flips"] >= 0 and p["regime_boundary_points_1e5"] > 0 and p["tw_over_unexplained"] == 0 and p["windows"]["ok"], p
    assert len(roof["frac_sets"]) == 3 and roof["frac_min"] <= roof["frac_median"] <= roof["frac_max"] and roof["frac_sets"][0] == roof["frac"], roof
    e = d["end_to_end"]
    assert e["points"] == 8 * 1800 * 3600 and e["arrays_in"] == 3 and e["arrays_out"] == 6, e
    assert 0 < e["kernel_ms"] < e["h2d_ms"] < e["d2h_ms"] and e["call_ms"] < e["phases_sum_ms"] * 1.5 and 10 < e["gbs"] < 120, e


def test_concurrent_calls_from_several_threads(ek):
    """The reference's functions are stateless and may be called from several threads; here every thread shares the
    block cache, the default stream and the library's device table.  Results must equal the serial ones."""
    import threading

    t, q, p = _fields(4, 1 << 18)
    jobs = [("pipeline_full", (t, q, p), {}), ("potential_temperature", (t, p), {}),
            ("wet_bulb_temperature_from_specific_humidity", (t, q, p), {"t_method": "newton"}),
            ("saturation_vapour_pressure", (t,), {"phase": "mixed"}), ("pipeline_svp_td_rh", (t, q, p), {}),
            ("wet_bulb_temperature_from_specific_humidity", (t, q, p), {})]
    want = [getattr(ek.thermo, f)(*a, **k) for f, a, k in jobs]
    got, errors = [[None] * 6 for _ in jobs], []

    def work(i):
        f, a, k = jobs[i]
        try:
            for rep in range(6):
                got[i][rep] = getattr(ek.thermo, f)(*a, **k)
        except BaseException as exc:
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for w, reps in zip(want, got):
        for g in reps:
            for a, b in zip(w if isinstance(w, tuple) else (w,), g if isinstance(g, tuple) else (g,)):
                assert np.array_equal(a, b, equal_nan=True)

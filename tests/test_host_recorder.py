"""Host logic that needs no GPU, with a RECORDER in place of libekm_thermo.so (every call logged, handles from a counter):
 * ekm_hip.graph(): the order of the runtime calls, which blocks a recording pins and when it lets go of them, and that
   everything that cannot be recorded raises before HIP is reached (tests/test_gpu_graph.py runs the real thing);
 * remembered plans: a device-resident call is planned once and repeats hand the library identical arguments;
 * the pinned result pool's eviction rule."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]

import ekm_hip  # noqa: E402
from ekm_hip import _ffi, device  # noqa: E402


class Recorder:
    """Stands in for libekm_thermo.so: every function returns 0 and is logged; handles come from a counter."""

    def __init__(self):
        self.calls, self.operands, self.next = [], [], 0x1000
        self.fail_end = False

    def _handle(self):
        self.next += 0x100
        return self.next

    def __getattr__(self, name):
        def fn(*args):
            plain = tuple(a for a in args if isinstance(a, (int, float, type(None))))
            self.calls.append((name,) + plain)
            self.operands.append([(o._obj.data, o._obj.mode, o._obj.len, o._obj.inner) for o in args
                                  if hasattr(o, "_obj") and isinstance(o._obj, _ffi.Operand)])
            if name in ("ekm_stream_create", "ekm_event_create", "ekm_malloc", "ekm_host_alloc"):
                args[-1]._obj.value = self._handle()
            if name == "ekm_graph_end":
                if self.fail_end:
                    return -3
                if args[-1] is not None:
                    args[-1]._obj.value = self._handle()
            if name == "ekm_last_error":
                return b"graph_end: the capture is invalid"
            return 0

        return fn

    def names(self):
        return [c[0] for c in self.calls]


@pytest.fixture
def rec(monkeypatch):
    r = Recorder()
    monkeypatch.setattr(_ffi, "lib", lambda: r)
    monkeypatch.setattr(device._tls, "device", 0, raising=False)
    monkeypatch.setattr(device._tls, "stream", None, raising=False)
    monkeypatch.setattr(device, "_cache", device._BlockCache())
    yield r
    device._tls.capture = None


def test_recording_brackets_and_replay(rec):
    t = ekm_hip.DeviceArray.empty((4, 8), np.float32)
    p = ekm_hip.DeviceArray.empty((4, 8), np.float32)
    rec.calls.clear()
    with ekm_hip.graph() as g:
        assert ekm_hip.current_stream() == g.stream and g.stream is not None
        th = ekm_hip.thermo.potential_temperature(t, p)
    assert ekm_hip.current_stream() is None
    assert isinstance(th, ekm_hip.DeviceArray) and th.shape == (4, 8)
    n = rec.names()
    # tables first, the device idle, a stream of its own, then the bracket around exactly one kernel launch
    assert n[:4] == ["ekm_prepare_tables", "ekm_sync", "ekm_stream_create", "ekm_graph_begin"]
    assert n[-1] == "ekm_graph_end" and n.count("ekm_potential_temperature_f32") == 1
    assert not any(c.startswith("ekm_event") or c in ("ekm_h2d", "ekm_d2h", "ekm_stream_sync") for c in n[4:-1]), n
    # the graph holds the two inputs and the result
    assert {id(a) for a in g._allocs} == {id(t._alloc), id(p._alloc), id(th._alloc)}
    assert t._alloc.pins == p._alloc.pins == th._alloc.pins == 1 and th._alloc.stream == g.stream

    # an input refreshed on the default stream: the launch orders the graph's stream after it (event record + wait)
    t._alloc.touch(None)
    rec.calls.clear()
    g.launch()
    n = rec.names()
    assert n[-1] == "ekm_graph_launch" and n.count("ekm_event_record") == 1 and n.count("ekm_stream_wait_event") == 1
    assert t._alloc.stream == g.stream
    rec.calls.clear()
    g.launch()
    assert rec.names() == ["ekm_graph_launch"] and g.launches == 2

    # a block the user frees while the graph lives stays allocated until the graph is closed
    th.free()
    assert th._alloc.free_pending and th._alloc.ptr
    rec.calls.clear()
    stream = g.stream
    g.close()
    n = rec.names()
    assert n[0] == "ekm_stream_sync" and "ekm_graph_destroy" in n and n[-1] == "ekm_stream_destroy"
    assert th._alloc.ptr is None and t._alloc.pins == 0 and t._alloc.ptr and t._alloc.stream is None
    assert ("ekm_stream_destroy", 0, stream) in rec.calls
    g.close()  # idempotent
    with pytest.raises(ekm_hip.EkmError, match="nothing recorded"):
        g.launch()


def test_what_cannot_be_recorded_raises_before_hip(rec):
    t = ekm_hip.DeviceArray.empty((16,), np.float32)
    host = np.ones(16, np.float32)
    with ekm_hip.graph() as g:
        rec.calls.clear()
        th = ekm_hip.thermo.potential_temperature(t, 85000.0)    # a Python scalar is recorded as a fill of its bit pattern
        assert isinstance(th, ekm_hip.DeviceArray)
        fills = [c for c in rec.calls if c[0] == "ekm_fill_u32"]
        assert len(fills) == 1 and fills[0][3] == int(np.float32(85000.0).view(np.uint32)) and fills[0][-1] == g.stream
        for bad in (lambda: ekm_hip.thermo.potential_temperature(host, host),       # NumPy operands are uploads
                    lambda: t.copy_from_host(host), lambda: t.to_host(), lambda: ekm_hip.to_device(host),
                    lambda: ekm_hip.synchronize(), lambda: np.asarray(t)):
            with pytest.raises(ekm_hip.EkmError, match=r"inside an ekm_hip.graph\(\) block"):
                bad()
        with pytest.raises(ekm_hip.EkmError, match="do not nest"):
            ekm_hip.graph().__enter__()
        assert not any(c in ("ekm_h2d", "ekm_d2h", "ekm_sync", "ekm_stream_sync") for c in rec.names()), rec.names()
    assert g._exec is not None
    with pytest.raises(ekm_hip.EkmError, match="recorded already"):
        g.__enter__()
    g.close()


def test_foreign_arrays_are_refused_while_recording(rec):
    class Foreign:
        def __dlpack_device__(self):
            return (10, 0)

        def __dlpack__(self, stream=None):
            raise AssertionError("handed a recording stream")

    with ekm_hip.graph() as g:
        with pytest.raises(ekm_hip.EkmError, match="from_dlpack before the block"):
            ekm_hip.thermo.saturation_vapour_pressure(Foreign())
        with pytest.raises(ekm_hip.EkmError, match=r"inside an ekm_hip.graph\(\) block"):
            ekm_hip.from_dlpack(Foreign())
    g.close()


def test_an_exception_in_the_block_ends_the_recording_and_releases_everything(rec):
    t = ekm_hip.DeviceArray.empty((16,), np.float32)
    with pytest.raises(RuntimeError, match="boom"):
        with ekm_hip.graph() as g:
            es = ekm_hip.thermo.saturation_vapour_pressure(t)
            raise RuntimeError("boom")
    ends = [c for c in rec.calls if c[0] == "ekm_graph_end"]
    assert len(ends) == 1 and ends[0][-1] is None      # recording dropped: no executable graph asked for
    assert g.stream is None and g._exec is None and t._alloc.pins == 0 and es._alloc.pins == 0
    assert ekm_hip.current_stream() is None and device._capturing() is None


def test_an_invalid_recording_raises_with_the_library_message(rec):
    t = ekm_hip.DeviceArray.empty((16,), np.float32)
    rec.fail_end = True
    with pytest.raises(ekm_hip.EkmError, match="capture is invalid"):
        with ekm_hip.graph() as g:
            ekm_hip.thermo.saturation_vapour_pressure(t)
    assert g.stream is None and t._alloc.pins == 0 and device._capturing() is None


def test_a_device_resident_call_is_planned_once_and_launched_from_the_recipe(rec, monkeypatch):
    """_engine.run: the first call with DeviceArray operands of given shapes / dtypes leaves a recipe behind; the next one
    skips the planning and must hand the library exactly the same arguments (new result pointers aside)."""
    from ekm_hip import _engine

    monkeypatch.setattr(_engine, "_recipes", {})
    t = ekm_hip.DeviceArray.empty((5, 6, 64), np.float32)
    q = ekm_hip.DeviceArray.empty((5, 6, 64), np.float32)
    lev = ekm_hip.DeviceArray.empty((5, 1, 1), np.float32)      # a level vector: LEVEL_MAJOR operand
    one = ekm_hip.DeviceArray.empty((), np.float32)             # a scalar operand
    planned = []
    real_plan = _engine._Plan
    monkeypatch.setattr(_engine, "_Plan", lambda *a, **k: planned.append(1) or real_plan(*a, **k))
    for args, call in (((t, q, lev), lambda: ekm_hip.thermo.wet_bulb_temperature_from_specific_humidity(t, q, lev, ept_method="bolton39", t_method="newton")),
                       ((t, one), lambda: ekm_hip.thermo.potential_temperature(t, one)),
                       ((t, q, lev), lambda: ekm_hip.thermo.saturation_mixing_ratio_slope(t, lev, eps=2e-4) if False else ekm_hip.thermo.pipeline_full(t, q, lev))):
        planned.clear()
        rec.calls.clear()
        rec.operands.clear()
        first, second, third = call(), call(), call()
        assert len(planned) == 1, planned                      # planned once, launched three times
        launches = [(c, o) for c, o in zip(rec.calls, rec.operands) if c[0].endswith("_f32") and "malloc" not in c[0]]
        assert len(launches) == 3
        nres = len(first) if isinstance(first, tuple) else 1
        strip = lambda c: c[:len(c) - nres - 1] + c[-1:]       # noqa: E731  (result pointers differ from call to call)
        assert strip(launches[0][0]) == strip(launches[1][0]) == strip(launches[2][0]), launches
        assert launches[0][1] == launches[1][1] == launches[2][1] and len(launches[0][1]) == len(args)
        for a, b in zip(first if isinstance(first, tuple) else (first,), second if isinstance(second, tuple) else (second,)):
            assert a.shape == b.shape == (5, 6, 64) and a.dtype == b.dtype == np.float32 and a.ptr != b.ptr
    # a different shape, dtype or enum argument is a different plan; NumPy operands never take the recipe
    planned.clear()
    ekm_hip.thermo.potential_temperature(t.reshape(30, 64), one)
    ekm_hip.thermo.wet_bulb_temperature_from_specific_humidity(t, q, lev, ept_method="ifs", t_method="newton")
    assert len(planned) == 2
    monkeypatch.setattr(device._tls, "devices", (0, 1), raising=False)   # as inside ekm_hip.multi_gpu([0, 1])
    with pytest.raises(ekm_hip.EkmError, match="multi_gpu"):
        ekm_hip.thermo.potential_temperature(t, one)            # remembered, but sharding goes the general way (and refuses)


def test_a_python_scalar_beside_device_arrays_is_uploaded_once(rec, monkeypatch):
    """`potential_temperature(t_dev, 85000.0)`: the scalar's value is kept on the device (one 0-d array per device, dtype
    and bit pattern) and the call becomes a remembered call on DeviceArrays -- the first call fills and plans, repeats
    hand the library one launch each; another value is another fill; recording a graph does not use the memory (the fill
    is a node of the graph there)."""
    from ekm_hip import _engine

    monkeypatch.setattr(_engine, "_recipes", {})
    monkeypatch.setattr(_engine, "_scalar_cache", type(_engine._scalar_cache)())
    t = ekm_hip.DeviceArray.empty((30, 64), np.float32)
    rec.calls.clear()
    a = ekm_hip.thermo.potential_temperature(t, 85000.0)
    first = rec.names()
    assert first.count("ekm_fill_u32") == 1 and first.count("ekm_potential_temperature_f32") == 1
    assert isinstance(a, ekm_hip.DeviceArray) and a.dtype == np.float32 and a.shape == (30, 64)
    rec.calls.clear()
    rec.operands.clear()
    for _ in range(3):
        ekm_hip.thermo.potential_temperature(t, 85000.0)
    assert [n for n in rec.names() if not n.startswith("ekm_malloc")] == ["ekm_potential_temperature_f32"] * 3   # no fill, no plan
    launches = [o for o in rec.operands if o]
    assert all(o == launches[0] for o in launches) and launches[0][1][1] == _ffi.SCALAR          # the same scalar block each time
    rec.calls.clear()
    ekm_hip.thermo.potential_temperature(t, 85000)              # an int of the same value: the same float32 bits
    ekm_hip.thermo.potential_temperature(t, 90000.0)            # another value: one more fill
    assert rec.names().count("ekm_fill_u32") == 1
    t64 = ekm_hip.DeviceArray.empty((30, 64), np.float64)
    rec.calls.clear()
    b = ekm_hip.thermo.potential_temperature(t64, 85000.0)      # weak scalar: the arrays' dtype wins, two words filled
    assert b.dtype == np.float64 and rec.names().count("ekm_fill_u32") == 2
    before = dict(_engine._scalar_cache)
    with ekm_hip.graph():
        ekm_hip.thermo.potential_temperature(t, 77000.0)        # recorded: the fill belongs to the graph, nothing remembered
    assert _engine._scalar_cache == before
    ekm_hip.empty_cache()
    assert _engine._scalar_cache == {}


def test_a_tiny_numpy_call_is_a_launch_and_a_wait(rec, monkeypatch):
    """The reference's quick start (README.md:39-58), two elements in, two out: operands written into one pinned block per
    thread, which the kernel reads and writes in place -- no device block, no copy in either direction; planned once."""
    from ekm_hip import _engine

    class Block:  # (the recorder hands out numbers, not memory: a real buffer in place of the pinned block)
        def __init__(self):
            _ffi.lib().ekm_host_alloc(_engine._TINY_BYTES, C.byref(C.c_void_p()))
            self.buf = C.create_string_buffer(_engine._TINY_BYTES)
            self.ptr = C.addressof(self.buf)

    monkeypatch.setattr(_engine, "_TinyBlock", Block)
    monkeypatch.setattr(_engine, "_tiny_recipes", {})
    monkeypatch.setattr(_engine, "_tiny_tls", type("T", (), {})())
    planned = []
    real = _engine._tiny_plan
    monkeypatch.setattr(_engine, "_tiny_plan", lambda *a: planned.append(a[0]) or real(*a))
    t, p = np.array([264.12, 261.45]), np.array([85000.0, 85000.0])
    rec.calls.clear()
    out = ekm_hip.thermo.potential_temperature(t, p)
    assert isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (2,)
    assert rec.names() == ["ekm_host_alloc", "ekm_potential_temperature_f64", "ekm_stream_sync"]
    rec.calls.clear()
    rec.operands.clear()
    for _ in range(3):
        again = ekm_hip.thermo.potential_temperature(t, p)
    assert rec.names() == ["ekm_potential_temperature_f64", "ekm_stream_sync"] * 3 and planned == ["potential_temperature"]
    assert again is not out and again.base is None                       # fresh arrays every time
    ops = [o for o in rec.operands if o]
    assert all(o == ops[0] for o in ops) and [m for _, m, _, _ in ops[0]] == [_ffi.FIELD, _ffi.FIELD]
    blk = _engine._tiny_tls.blocks[0]
    assert np.frombuffer(blk.buf, np.float64, 2, ops[0][0][0] - blk.ptr).tolist() == t.tolist()   # the operands sit in the block
    # scalars in, a NumPy scalar out; a Python scalar beside float32 arrays is weak where the reference uses it as it comes
    # (the result stays float32) and float64 where the reference passes it through asarray first (potential_temperature:
    # recorded from the reference, ekm_hip/_dtype_rules.py); float16 computes in float32 and comes back as float16
    s = ekm_hip.thermo.potential_temperature(264.12, 85000.0)
    assert isinstance(s, np.float64)
    t32 = t.astype(np.float32)
    assert ekm_hip.thermo.relative_humidity_from_specific_humidity(t32, np.full(2, 0.004, np.float32), 85000.0).dtype == np.float32
    assert ekm_hip.thermo.potential_temperature(t32, 85000.0).dtype == np.float64
    assert ekm_hip.thermo.potential_temperature(t.astype(np.float16), (p * 0.5).astype(np.float16)).dtype == np.float16
    # a level vector against a small field keeps its broadcast class; lists and big arrays go the general way
    rec.calls.clear()
    ekm_hip.thermo.potential_temperature(np.ones((4, 16)), np.linspace(5e4, 1e5, 4)[:, None])
    assert rec.names()[-2:] == ["ekm_potential_temperature_f64", "ekm_stream_sync"] and rec.operands[-2][1][1] == _ffi.LEVEL_MAJOR
    rec.calls.clear()
    ekm_hip.thermo.potential_temperature([264.12, 261.45], [85000, 85000])
    assert "ekm_h2d" in rec.names() and "ekm_d2h" in rec.names()
    rec.calls.clear()
    ekm_hip.thermo.potential_temperature(np.ones(1 << 16), np.ones(1 << 16))
    assert "ekm_h2d" in rec.names()


def test_pinned_pool_lets_go_of_the_sizes_that_went_unused_longest(rec, monkeypatch):
    """device._PinnedPool: ONE bound for all its page-locked memory (blocks handed out + blocks cached <= limit).  A cache
    full of one block size must not keep a workload with a new field size from getting pinned blocks: the oldest-unused
    sizes are let go to make room; what does not fit is refused (the caller takes a pageable array)."""
    pool = device._PinnedPool()
    pool.limit, pool.live_limit = 8 << 20, 8 << 20
    rec.calls.clear()
    small = [pool.take(2 << 20) for _ in range(4)]              # four 2-MiB blocks: allocated
    assert [c[0] for c in rec.calls] == ["ekm_host_alloc"] * 4
    assert pool.take(2 << 20)[0] is None                        # a fifth does not fit the bound: refused, nothing pinned
    for ptr, b in small:
        pool.give(ptr, b)
    assert pool.cached == 8 << 20 and pool.handed_out == 0      # all four cached: the pool is full
    rec.calls.clear()
    big, bb = pool.take(6 << 20)                                # a new size: three of the old blocks let go, then allocated
    assert [c[0] for c in rec.calls] == ["ekm_host_free", "ekm_host_free", "ekm_host_free", "ekm_host_alloc"]
    assert pool.handed_out + pool.cached == 8 << 20
    pool.give(big, bb)
    assert pool.cached == (6 << 20) + (2 << 20) and pool.handed_out == 0
    rec.calls.clear()
    again, _ = pool.take(6 << 20)                               # recycled: no allocation
    assert again == big and rec.calls == []
    pool.give(again, bb)
    huge, hb = pool.take(16 << 20)                              # larger than the whole bound: never pinned
    assert huge is None and pool.handed_out == 0
    pool.live_limit = 4 << 20                                   # callers may hold less than the pool may cache
    assert pool.take(6 << 20)[0] is None and pool.take(2 << 20)[0] is not None
    pool.drain()
    assert pool.cached == 0


def test_pinned_pool_bound_comes_from_the_machine(monkeypatch):
    """min(25 % of MemAvailable, 16 GiB) unless EKM_PINNED_CACHE_BYTES says otherwise; memory_stats() reports it."""
    monkeypatch.delenv("EKM_PINNED_CACHE_BYTES", raising=False)
    monkeypatch.delenv("EKM_PINNED_LIVE_BYTES", raising=False)
    pool = device._PinnedPool()
    assert 0 < pool.limit <= 16 << 30 and pool.live_limit == pool.limit
    with open("/proc/meminfo") as f:
        avail = next(int(ln.split()[1]) * 1024 for ln in f if ln.startswith("MemAvailable:"))
    assert pool.limit <= avail // 4 + (1 << 30)
    monkeypatch.setenv("EKM_PINNED_CACHE_BYTES", str(3 << 20))
    assert device._PinnedPool().limit == 3 << 20

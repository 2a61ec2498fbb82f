"""The gfx950 kernels on the wide fuzz domain of tests/_fuzz.py, through the C ABI, against the oracle; and the tree
walk's identity: the default walk (sign tests without a transcendental) against the same walk with the reference's own
residual at every step (tuning parameter bisect_exact), bit for bit -- on the fuzz domain and on eight levels of the
benchmark field, every theta_e method, both precisions.  (Round 4 kept that comparison in a builder tool,
tools/bisect_equiv.py, and ran it only where the defects VERDICT r4 found cannot occur.)"""
import numpy as np
import pytest

import _fuzz

pytestmark = pytest.mark.gpu
np.seterr(all="ignore")
INNER = 1800 * 3600
SEED = 20260313


@pytest.fixture(scope="module", params=["f32", "f64"])
def points(request):
    dtype = np.float32 if request.param == "f32" else np.float64
    return request.param, dtype, _fuzz.make(dtype=dtype)


@pytest.fixture(scope="module")
def dev_points(ek, points):
    tag, dtype, d = points
    dd = {k: ek.to_device(v) for k, v in d.items()}
    yield dd
    for a in dd.values():
        a.free()


@pytest.mark.parametrize("func,keys,method,t_method", _fuzz.CASES,
                         ids=[f"{f.split('_')[0]}-{m}-{tm}" for f, _, m, tm in _fuzz.CASES])
def test_fuzz_against_the_oracle(ek, points, dev_points, func, keys, method, t_method):
    tag, dtype, d = points
    out = getattr(ek.thermo, func)(*[dev_points[k] for k in keys], ept_method=method, t_method=t_method)
    got = out.to_host()
    out.free()
    print(_fuzz.judge(func, keys, method, t_method, tag, d, got))


@pytest.fixture(scope="module", params=["f32", "f64"])
def adversarial(request):
    dtype = np.float32 if request.param == "f32" else np.float64
    return request.param, dtype, _fuzz.make(n=_fuzz.MORE_POINTS, seed=_fuzz.SEED + 1, dtype=dtype, adversarial=True)


@pytest.mark.parametrize("func,keys,method,t_method", _fuzz.CASES, ids=[f"{f.split('_')[0]}-{m}-{tm}" for f, _, m, tm in _fuzz.CASES])
def test_fuzz_next_to_the_node_pressures_p0_and_saturation(ek, adversarial, func, keys, method, t_method):
    """_fuzz.make(adversarial=True): 60 % of the points within 1e-2 ... 1e-7 of es at one of the tree's first 127 nodes, within
    1e-3 ... 1e-7 of p0, or saturated (td = t)."""
    tag, dtype, d = adversarial
    got = getattr(ek.thermo, func)(*[d[k] for k in keys], ept_method=method, t_method=t_method)
    print(_fuzz.judge(func, keys, method, t_method, tag, d, got))


@pytest.mark.parametrize("func,keys,method,t_method", _fuzz.CASES_MORE,
                         ids=[f"{'-'.join(f.split('_')[:3] + f.split('_')[-1:])}-{m}-{tm}" for f, _, m, tm in _fuzz.CASES_MORE])
def test_fuzz_of_the_other_callers_of_the_inversions(ek, points, func, keys, method, t_method):
    """theta_e from the dewpoint and theta_w (the inversion at p0) -- NumPy in, NumPy out: the streamed host path."""
    tag, dtype, d = points
    d = _fuzz.head(d)
    got = getattr(ek.thermo, func)(*[d[k] for k in keys], ept_method=method, t_method=t_method)
    assert isinstance(got, np.ndarray)
    print(_fuzz.judge(func, keys, method, t_method, tag, d, got))


@pytest.mark.parametrize("func,keys,kwargs", _fuzz.DIRECT, ids=[f"{f}-{'-'.join(map(str, kw.values()))}" for f, _, kw in _fuzz.DIRECT])
def test_direct_functions_on_the_fuzz_domain(ek, points, dev_points, func, keys, kwargs):
    tag, dtype, d = points
    out = getattr(ek.thermo, func)(*[dev_points[k] for k in keys], **kwargs)
    got = out.to_host()
    out.free()
    print(_fuzz.judge_direct(func, keys, kwargs, tag, d, got))


@pytest.mark.parametrize("func,keys,kwargs", _fuzz.DIRECT_REST, ids=[f"{f}-{'-'.join(map(str, kw.values()))}" for f, _, kw in _fuzz.DIRECT_REST])
def test_every_other_closed_form_case_on_the_fuzz_domain(ek, points, func, keys, kwargs):
    """The rest of the 39 functions x variants (NumPy in, NumPy out)."""
    tag, dtype, d = points
    d = _fuzz.head(d)
    got = getattr(ek.thermo, func)(*[d[k] for k in keys], **kwargs)
    print(_fuzz.judge_direct(func, keys, kwargs, tag, d, got))


@pytest.mark.parametrize("name", sorted(_fuzz.FUSED))
def test_fused_pipelines_on_the_fuzz_domain(ek, points, dev_points, name):
    """BASELINE configs 3 and 5 far outside the benchmark distribution: every output held to the separate function's bar."""
    tag, dtype, d = points
    outs = getattr(ek.thermo, name)(*[dev_points[k] for k in ("t", "q", "p")])
    got = [o.to_host() for o in outs]
    for o in outs:
        o.free()
    print(_fuzz.judge_fused(name, tag, d, got))


LEVELS = 64


@pytest.mark.parametrize("method,t_method", [(m, tm) for m in _fuzz.METHODS for tm in _fuzz.T_METHODS] + [("ifs", "fused")])
def test_level_vector_pressure_on_the_fuzz_domain(ek, points, method, t_method):
    """The same draws of t and q as 64 levels x 16384 points with the pressure a LEVEL VECTOR (64 log-spaced levels, 1 Pa ...
    1.26e5 Pa): the kernels that take p as a wave-uniform value per level (map_levels) on the wide domain -- the wet-bulb
    by every method and the six-output pipeline, judged as the field-mode results are.  (N-d input to the bisection is a
    superset of the reference, whose own bisection takes 1-d input only.)"""
    tag, dtype, d = points
    n = d["t"].size // LEVELS
    t, q = d["t"].reshape(LEVELS, n), d["q"].reshape(LEVELS, n)
    p = np.exp(np.linspace(np.log(1.0), np.log(1.26e5), LEVELS)).astype(dtype).reshape(LEVELS, 1)
    flat = dict(t=t.ravel(), q=q.ravel(), p=np.broadcast_to(p, t.shape).ravel().copy())
    dt_, dq, dp = ek.to_device(t), ek.to_device(q), ek.to_device(p)
    try:
        if t_method == "fused":
            outs = ek.thermo.pipeline_full(dt_, dq, dp)
            got = [o.to_host().ravel() for o in outs]
            for o in outs:
                o.free()
            print(_fuzz.judge_fused("pipeline_full", tag, flat, got))
        else:
            out = ek.thermo.wet_bulb_temperature_from_specific_humidity(dt_, dq, dp, ept_method=method, t_method=t_method)
            got = out.to_host().ravel()
            out.free()
            print(_fuzz.judge("wet_bulb_temperature_from_specific_humidity", ("t", "q", "p"), method, t_method, tag, flat, got))
    finally:
        for a in (dt_, dq, dp):
            a.free()


@pytest.mark.parametrize("method,t_method", [("ifs", "bisect"), ("ifs", "newton"), ("bolton35", "bisect"), ("bolton39", "newton"), ("ifs", "fused")])
def test_hybrid_level_pressure_on_the_fuzz_domain(ek, points, method, t_method):
    """The same draws of t and q on the 137 IFS hybrid levels, the pressure formed INSIDE the kernels from a surface pressure
    of 500 ... 1100 hPa (HybridPressure: bands of hybrid levels + the pure pressure levels as a level-vector launch)."""
    from oracle import vertical_oracle as vo

    tag, dtype, d = points
    nlev = 137
    n = (d["t"].size // nlev) // 8 * 8
    t, q = d["t"][:nlev * n].reshape(nlev, n), d["q"][:nlev * n].reshape(nlev, n)
    A, B = (x.astype(dtype) for x in ek.vertical.hybrid_level_parameters(nlev))
    sp = np.random.default_rng(7).uniform(5e4, 1.1e5, n).astype(dtype)
    p = np.stack([vo.pressure_on_hybrid_levels(A[k:k + 2], B[k:k + 2], sp)[0].reshape(n) for k in range(nlev)]).astype(dtype)
    flat = dict(t=t.ravel(), q=q.ravel(), p=p.ravel())
    dt_, dq, dsp = ek.to_device(t), ek.to_device(q), ek.to_device(sp)
    hp = ek.HybridPressure(A, B, dsp)
    try:
        if t_method == "fused":
            outs = ek.thermo.pipeline_full(dt_, dq, hp)
            got = [o.to_host().ravel() for o in outs]
            for o in outs:
                o.free()
            print(_fuzz.judge_fused("pipeline_full", tag, flat, got))
        else:
            out = ek.thermo.wet_bulb_temperature_from_specific_humidity(dt_, dq, hp, ept_method=method, t_method=t_method)
            got = out.to_host().ravel()
            out.free()
            print(_fuzz.judge("wet_bulb_temperature_from_specific_humidity", ("t", "q", "p"), method, t_method, tag, flat, got))
    finally:
        for a in (dt_, dq, dsp):
            a.free()


def _exact(on):
    from ekm_hip import _ffi

    _ffi.check(_ffi.lib().ekm_set_tuning_param(b"bisect_exact", 1 if on else 0))


def _same_bits(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("method", _fuzz.METHODS)
def test_default_walk_is_the_exact_walk_on_the_fuzz_domain(ek, points, dev_points, method):
    tag, dtype, d = points
    try:
        for func, keys in _fuzz.FUNCS + _fuzz.FUNCS_MORE:
            ins = [dev_points[k] for k in keys]
            _exact(False)
            fast = getattr(ek.thermo, func)(*ins, ept_method=method, t_method="bisect")
            _exact(True)
            exact = getattr(ek.thermo, func)(*ins, ept_method=method, t_method="bisect")
            a, b = fast.to_host(), exact.to_host()
            fast.free()
            exact.free()
            diff = ~_same_bits(a, b)
            assert not diff.any(), (f"{func}[{method},{tag}]: {int(diff.sum())} points differ between the default and the exact "
                                    f"walk, e.g. {np.flatnonzero(diff)[:4]}: {a[diff][:4]} vs {b[diff][:4]}")
    finally:
        _exact(False)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("method", _fuzz.METHODS)
def test_default_walk_is_the_exact_walk_on_eight_levels_of_the_benchmark_field(ek, method, tag):
    """Levels 0, 20, ..., 136 of the 137 (the generator works on the global point index: any level can be filled alone),
    pressure as a full field, as the level vector and on hybrid levels is covered by tools/bisect_equiv.py; here the
    default operand mode of the benchmark."""
    from ekm_hip import _ffi

    dtype = np.float32 if tag == "f32" else np.float64
    lib = _ffi.lib()
    fill = getattr(lib, f"ekm_synth_fill_{tag}")
    t, q, p = (ek.DeviceArray.empty((INNER,), dtype) for _ in range(3))
    total = 0
    try:
        for lev in (0, 20, 40, 60, 80, 100, 120, 136):
            _ffi.check(fill(0, None, t.ptr, q.ptr, p.ptr, lev * INNER, INNER, INNER, 137, SEED))
            ek.synchronize()
            _exact(False)
            fast = ek.thermo.wet_bulb_temperature_from_specific_humidity(t, q, p, ept_method=method, t_method="bisect")
            _exact(True)
            exact = ek.thermo.wet_bulb_temperature_from_specific_humidity(t, q, p, ept_method=method, t_method="bisect")
            a, b = fast.to_host(), exact.to_host()
            fast.free()
            exact.free()
            diff = ~_same_bits(a, b)
            assert not diff.any(), f"level {lev} [{method},{tag}]: {int(diff.sum())} of {INNER} points differ, e.g. {np.flatnonzero(diff)[:4]}"
            assert np.isfinite(a).mean() > 0.99
            total += INNER
    finally:
        _exact(False)
        for a in (t, q, p):
            a.free()
    assert total == 8 * INNER


@pytest.mark.parametrize("dtype,t,q,p,expect", __import__("test_hosttwin_fuzz").B35_UNDERFLOW)
def test_bolton35_stays_on_the_node_where_both_terms_underflow(ek, dtype, t, q, p, expect):
    got = ek.thermo.wet_bulb_temperature_from_specific_humidity(np.array([t], dtype), np.array([q], dtype), np.array([p], dtype),
                                                                ept_method="bolton35", t_method="bisect")
    assert got.dtype == dtype and abs(float(got[0]) - expect) < 1e-4, got


@pytest.mark.parametrize("dtype,t,td,p,expect", __import__("test_hosttwin_fuzz").B39_INFINITE_EPT)
def test_bolton39_with_an_infinite_theta_e(ek, dtype, t, td, p, expect):
    got = ek.thermo.wet_bulb_temperature_from_dewpoint(np.array([t], dtype), np.array([td], dtype), np.array([p], dtype),
                                                       ept_method="bolton39", t_method="bisect")
    assert got.dtype == dtype and abs(float(got[0]) - expect) < 1e-4, got


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("func,keys,kwargs", _fuzz._case_table(), ids=[f"{f}-{'-'.join(map(str, kw.values()))}" for f, _, kw in _fuzz._case_table()])
def test_special_operands_in_every_combination(ek, tag, func, keys, kwargs):
    """NaN, infinities, zeros, negatives, 1e-30 and 1e30 in every combination, all 39 functions x variants (NumPy in / out)."""
    dtype = np.float32 if tag == "f32" else np.float64
    ins = _fuzz.special_operands(keys, dtype)
    got = getattr(ek.thermo, func)(*ins, **kwargs)
    print(_fuzz.judge_special(func, keys, kwargs, tag, ins, got))


def test_random_shapes_dtypes_and_layouts_through_the_numpy_path(ek):
    """tools/shape_fuzz.py: 300 random calls -- broadcast patterns, float32 / float64 / Python-scalar / list / 0-d operands,
    Fortran-ordered, strided and reversed views -- against the oracle, which agrees with the reference itself on all of them
    (`--reference` in the build container: 0 differences): result shape, result dtype (ekm_hip/_dtype_rules.py), values."""
    import importlib.util
    import os
    import sys

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "shape_fuzz.py")
    spec = importlib.util.spec_from_file_location("shape_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["shape_fuzz.py", "--trials", "300"]
        assert mod.main() == 0
        sys.argv = ["shape_fuzz.py", "--trials", "1000", "--all", "--seed", "7"]  # drawn from all 94 function x variant cases
        assert mod.main() == 0
        sys.argv = ["shape_fuzz.py", "--trials", "300", "--device"]  # the same calls with DeviceArray operands: shapes and values
        assert mod.main() == 0
    finally:
        sys.argv = argv


@pytest.mark.parametrize("mode", [["--stream"], ["--multi", "3", "--seed", "2"]], ids=["streamed", "sharded"])
def test_random_shapes_through_the_streamed_and_the_sharded_path(mode):
    """tools/shape_fuzz.py with every NumPy call forced through the streamed path (slices of the leading axis, uploader and
    downloader threads, a 24-MiB budget) resp. inside multi_gpu([0, 0, 0]) -- in a child process: it lowers module-level
    thresholds of the engine."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "shape_fuzz.py"), "--trials", "300"] + mode, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "300 trials, 0 differences" in r.stdout, (r.stdout[-800:], r.stderr[-800:])


@pytest.mark.parametrize("kind", sorted(_fuzz.ILL_CONDITIONED))
def test_where_the_reference_is_ill_conditioned_we_miss_no_more_often_than_it_does(ek, kind):
    """Bolton-35 + Newton next to p = p0 and theta_w by Newton of stratospheric parcels -- physical input on
    which the reference's own fp32 and fp64 runs disagree beyond 1e-4 on tens of points per million.  The README's parity
    claim excepts exactly these; this test keeps the exception honest (tests/_fuzz.py::judge_vs_reference_spread)."""
    d = _fuzz.make_ill_conditioned(kind)
    func, keys, kwargs = _fuzz.ILL_CONDITIONED[kind]
    got = getattr(ek.thermo, func)(*[d[k] for k in keys], **kwargs)
    print(_fuzz.judge_vs_reference_spread(kind, d, got))

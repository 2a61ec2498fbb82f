"""pressure_on_hybrid_levels on the GPU (ekm_hip.vertical) against the recorded reference vectors.
fp64: 1e-6 relative (measured ~1e-15).  fp32: the reference's own fp32 tolerances for this function
(tests/vertical/test_array_vertical.py:192-198 there): p 1e-4 relative, delta atol 1e-6 rtol 1e-5,
alpha atol 1e-4 rtol 1e-5 (alpha = 1 - x*delta cancels to ~1 % of its terms)."""
import os

import numpy as np
import pytest

from test_vertical_oracle import CASES, G, case_args

pytestmark = pytest.mark.gpu


def _close(got, want, name, f32):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if not f32:
        return np.allclose(got, want, rtol=1e-6, atol=0)
    atol, rtol = {"full": (0, 1e-4), "half": (0, 1e-4), "delta": (1e-6, 1e-5), "alpha": (1e-4, 1e-5)}[name]
    return np.allclose(got, want, rtol=rtol, atol=atol)


@pytest.mark.parametrize("c", CASES, ids=[c["id"] for c in CASES])
def test_golden(ek, c):
    A, B, sp = case_args(c)
    res = ek.vertical.pressure_on_hybrid_levels(A, B, sp, levels=c["levels"], alpha_top=c["alpha_top"],
                                                output=c["output"], vertical_axis=c["vertical_axis"])
    res = res if isinstance(res, tuple) else (res,)
    for name, r, dt in zip(c["output"], res, c["out_dtype"]):
        w = G[f"{c['id']}.{name}"]
        assert str(r.dtype) == dt, (c["id"], name, r.dtype, dt)
        assert _close(r, w, name, c["dtype"] == "f32"), (c["id"], name, np.abs(r - w).max())


def test_reference_fixture(ek):
    A, B, sp = G["fixture.A"], G["fixture.B"], G["fixture.p_surf"]
    full, half, delta, alpha = ek.vertical.pressure_on_hybrid_levels(A, B, sp, output=["full", "half", "delta", "alpha"])
    for got, name in ((full, "p_full"), (half, "p_half"), (delta, "delta"), (alpha, "alpha")):
        assert np.allclose(got, G[f"fixture.{name}"], atol=1e-8, rtol=1e-6), name


def test_device_resident_and_errors(ek):
    from oracle import vertical_oracle as vo

    A, B = G["coef.137.A"], G["coef.137.B"]
    rng = np.random.default_rng(3)
    sp = rng.uniform(5e4, 1.05e5, (37, 53)).astype(np.float32)
    dsp = ek.to_device(sp)
    full, alpha = ek.vertical.pressure_on_hybrid_levels(A.astype(np.float32), B.astype(np.float32), dsp,
                                                        output=["full", "alpha"])
    assert isinstance(full, ek.DeviceArray) and full.shape == (137, 37, 53) and full.dtype == np.float32
    wf, wa = vo.pressure_on_hybrid_levels(A.astype(np.float32), B.astype(np.float32), sp, output=["full", "alpha"])
    assert _close(full.to_host(), wf, "full", True) and _close(alpha.to_host(), wa, "alpha", True)
    # a level range whose top half level is far from 0 Pa and has B != 0: the device-side any() test
    lv = [100, 101, 137]
    d = ek.vertical.pressure_on_hybrid_levels(A.astype(np.float32), B.astype(np.float32), dsp, levels=lv, output="delta")
    assert _close(d.to_host(), vo.pressure_on_hybrid_levels(A.astype(np.float32), B.astype(np.float32), sp, levels=lv,
                                                            output="delta"), "delta", True)
    with pytest.raises(ValueError, match="Unknown output type"):
        ek.vertical.pressure_on_hybrid_levels(A, B, sp, output="bogus")
    with pytest.raises(ValueError, match="exceeds the maximum"):
        ek.vertical.pressure_on_hybrid_levels(A, B, sp, levels=[138])
    with pytest.raises(ValueError, match="starts at 1"):
        ek.vertical.pressure_on_hybrid_levels(A, B, sp, levels=[0])


@pytest.mark.parametrize("tag,dt", [("f32", np.float32), ("f64", np.float64)])
def test_hybrid_pressure_operand(ek, tag, dt):
    """Thermo kernels fed the DEFINITION of the model-level pressure (A, B, sp) instead of a pressure
    field: same results as with the materialised field from pressure_on_hybrid_levels."""
    from _compare import assert_parity
    from oracle import synthetic, thermo_oracle as orc, vertical_oracle as vo

    # the lowest 101 layers (p > ~25 hPa): above that the synthetic humidity is unphysical and the
    # reference's own fp32 and fp64 wet-bulb disagree at the 1e-2 level
    A, B = G["coef.137.A"][36:], G["coef.137.B"][36:]
    rng = np.random.default_rng(11)
    for npts in (1031, 4096):  # odd: chunks straddle level boundaries; multiple of 4: the aligned path
        sp = rng.uniform(5.2e4, 1.04e5, npts).astype(dt)
        pfull = vo.pressure_on_hybrid_levels(A.astype(dt), B.astype(dt), sp)
        t = (synthetic.standard_temperature(pfull) + rng.normal(0, 8, pfull.shape)).astype(dt)
        q = np.minimum(orc.specific_humidity_from_relative_humidity(t.astype(np.float64), rng.uniform(1, 100, t.shape),
                                                                    pfull.astype(np.float64)), 0.04).astype(dt)
        hp = ek.HybridPressure(A, B, sp)
        for func, args in (("potential_temperature", (t,)), ("relative_humidity_from_specific_humidity", (t, q)),
                           ("pipeline_svp_td_rh", (t, q)), ("pipeline_full", (t, q))):
            got = getattr(ek.thermo, func)(*args, hp)
            want = getattr(orc, func)(*[a.copy() for a in args], pfull.copy())
            got = got if isinstance(got, tuple) else (got,)
            want = want if isinstance(want, tuple) else (want,)
            from oracle import conditioning
            edge = conditioning.newton_regime_boundary("pipeline_full", [t, q, pfull], {}, 1e-5 if tag == "f32" else 1e-13)
            for k, (g_, w_) in enumerate(zip(got, want)):
                assert g_.shape == pfull.shape and g_.dtype == dt
                assert_parity(g_, w_, tag, f"{func} hybrid p {tag} npts={npts}",
                              unstable=edge if (func == "pipeline_full" and k == 5) else None)
    # device-resident sp and fields
    dsp, dtt = ek.to_device(sp), ek.to_device(t)
    th = ek.thermo.potential_temperature(dtt, ek.HybridPressure(A, B, dsp))
    assert isinstance(th, ek.DeviceArray)
    assert_parity(th.to_host(), orc.potential_temperature(t, pfull), tag, "device-resident hybrid p")
    with pytest.raises(ValueError):
        ek.thermo.potential_temperature(t[:5], hp)


def test_hybrid_pressure_with_non_finite_surface_pressure(ek):
    """The pure pressure levels of a hybrid table (B = 0) run as a level-vector launch that does not need sp -- but the
    reference's p = A + B*sp is NaN there for a NaN or infinite sp (0*inf), and so must ours be, on every level, for
    exactly those columns (aligned and ragged shapes; the all-B-zero sub-table too)."""
    from oracle import thermo_oracle as orc, vertical_oracle as vo

    A, B = ek.vertical.hybrid_level_parameters(137)
    rng = np.random.default_rng(4)
    for npts, sub in ((4096, slice(0, 138)), (1031, slice(0, 138)), (2048, slice(0, 40)), (2048, slice(60, 138))):
        a_, b_ = A[sub], B[sub]
        nlev = a_.size - 1
        sp = rng.uniform(5.2e4, 1.04e5, npts).astype(np.float32)
        bad = rng.choice(npts, 7, replace=False)
        sp[bad[:3]], sp[bad[3:5]], sp[bad[5:]] = np.nan, np.inf, -np.inf
        pfull = vo.pressure_on_hybrid_levels(a_.astype(np.float32), b_.astype(np.float32), sp)
        t = rng.uniform(200.0, 300.0, (nlev, npts)).astype(np.float32)
        q = rng.uniform(1e-6, 5e-3, (nlev, npts)).astype(np.float32)
        hp = ek.HybridPressure(a_, b_, sp)
        flat = 0
        while flat < nlev and b_[flat] == 0 and b_[flat + 1] == 0:
            flat += 1
        assert hp.nflat == flat and flat == {0: min(53, nlev), 60: 0}[sub.start]
        for func, args, kw in (("potential_temperature", (t,), {}), ("pipeline_svp_td_rh", (t, q), {}),
                               ("wet_bulb_temperature_from_specific_humidity", (t, q), {"t_method": "newton"}),
                               ("wet_bulb_temperature_from_specific_humidity", (t, q), {"t_method": "bisect"})):
            got = getattr(ek.thermo, func)(*args, hp, **kw)
            if kw.get("t_method") == "bisect":  # the reference's bisection takes 1-D input only
                want = getattr(orc, func)(*(a.ravel() for a in args), pfull.ravel(), **kw).reshape(pfull.shape)
            else:
                want = getattr(orc, func)(*args, pfull, **kw)
            got = got if isinstance(got, tuple) else (got,)
            want = want if isinstance(want, tuple) else (want,)
            for k, (g_, w_) in enumerate(zip(got, want)):
                if func == "pipeline_svp_td_rh" and k == 0:
                    continue  # es depends on t alone
                assert np.array_equal(np.isnan(g_), np.isnan(w_)), (func, npts, sub, k)
                assert np.isnan(g_[:, bad]).all() and np.isfinite(g_[:, np.setdiff1d(np.arange(npts), bad)][nlev // 2:]).all()


from test_vertical_oracle import CHAIN, chain_args, chain_calls  # noqa: E402


@pytest.mark.parametrize("c", CHAIN, ids=[c["id"] for c in CHAIN])
def test_geopotential_chain_golden(ek, c):
    """Fused column scan (t, q, sp -> thickness / geopotential / heights) vs the reference's outputs.
    fp64: 1e-6 relative.  fp32: the reference's own fp32 tolerance for this chain (atol 10 m2/s2 resp.
    10/g m, rtol 1e-6; tests/vertical/test_array_vertical.py:430-431 there) against its fp32 result --
    alpha = 1 - x*log(r) of a thin layer cancels to 1e-3 of its terms, and the reference's fp32 evaluation
    of it is itself 2 % off there -- AND within 2e-4 relative (+0.05 absolute) of the reference's fp64
    result, which the kernel's cancellation-free series tracks more closely than the fp32 reference does."""
    args = chain_args(c)
    for k, v in chain_calls(ek.vertical, *args).items():
        w = G[f"{c['id']}.{k}"]
        assert v.dtype == w.dtype and v.shape == w.shape, (c["id"], k, v.dtype, w.dtype)
        if c["dtype"] == "f64":
            assert np.allclose(v, w, rtol=1e-6, atol=1e-9), (c["id"], k, np.abs(v - w).max())
        else:
            atol = 10.0 if k in ("thickness", "geopotential") else 10.0 / 9.80665
            assert np.allclose(v, w, rtol=1e-6, atol=atol), (c["id"], k, np.abs(v - w).max())
            w64 = G[f"{c['id'].replace('.f32.', '.f64.')}.{k}"]
            assert np.allclose(v, w64, rtol=2e-4, atol=0.05), (c["id"], k, np.abs(v - w64).max())


def test_geopotential_chain_fixtures_and_device(ek):
    A, B, sp, t, q = (G[f"fixture.{k}"] for k in ("A", "B", "p_surf", "t", "q"))
    z = ek.vertical.relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp)
    assert np.allclose(z, G["fixture.z"], atol=1e-8, rtol=1e-6)
    z = ek.vertical.geopotential_on_hybrid_levels(t[90:], q[90:], np.zeros(2), A, B, sp)
    assert np.allclose(z, G["fixture.z"][90:], atol=1e-8, rtol=1e-6)
    A137, B137 = G["coef.137.A"], G["coef.137.B"]
    sp, zs, t, q = (G[f"hfix.{k}"] for k in ("p_surf", "z_surf", "t", "q"))
    for ht in ("geometric", "geopotential"):
        for hr in ("sea", "ground"):
            h = ek.vertical.height_on_hybrid_levels(t, q, zs, A137, B137, sp, h_type=ht, h_reference=hr)
            assert np.allclose(h, G[f"hfix.h_{ht}_{hr}"], atol=1e-8, rtol=1e-6), (ht, hr)
    # device-resident, fp32, ragged column count
    from oracle import vertical_oracle as vo

    c = dict(dtype="f32", nlev=137)
    t, q, zs, A, B, sp = chain_args(c)
    out = ek.vertical.geopotential_on_hybrid_levels(ek.to_device(t), ek.to_device(q), ek.to_device(zs), A, B,
                                                    ek.to_device(sp))
    assert isinstance(out, ek.DeviceArray) and out.shape == t.shape
    assert np.allclose(out.to_host(), vo.geopotential_on_hybrid_levels(t, q, zs, A, B, sp), rtol=1e-6, atol=10.0)
    with pytest.raises(ValueError, match="h_reference"):
        ek.vertical.height_on_hybrid_levels(t, q, zs, A, B, sp, h_reference="moon")


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_thickness_from_alpha_delta(ek, dt):
    """vertical.py:815-893: the column scan with alpha and delta streamed instead of formed.  Inputs: the alpha / delta
    the reference itself was given when the vectors were recorded (fp64 whatever the input dtype)."""
    npdt = np.float32 if dt == "f32" else np.float64
    F = ek.vertical.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta
    t, q = (G[f"chain.{k}"].astype(npdt) for k in ("t", "q"))
    al, de = G[f"chain.{dt}.alpha"], G[f"chain.{dt}.delta"]
    got = F(t, q, al, de)
    w = G[f"chain.{dt}.from_alpha_delta"]
    assert got.dtype == w.dtype == npdt and got.shape == w.shape  # the result has the dtype of t and q (vertical.py:762)
    assert np.allclose(got, w, rtol=1e-6, atol=1e-9), np.abs(got - w).max()  # the arithmetic ran in fp64 (alpha, delta)
    # the same through the fused producer (A, B, sp): identical up to the fp64 bar
    if dt == "f64":
        fused = ek.vertical.relative_geopotential_thickness_on_hybrid_levels(t, q, G["coef.137.A"], G["coef.137.B"],
                                                                             G["chain.sp"])
        assert np.allclose(got, fused, rtol=1e-6, atol=1e-9)
    if dt == "f32":
        # all-fp32 call: the reference's own fp32 bar for this chain (atol 10 m2/s2, rtol 1e-6) vs its fp32 result
        g32 = F(t, q, al.astype(npdt), de.astype(npdt))
        w32 = G["chain.f32.from_alpha_delta_f32ad"]
        assert g32.dtype == np.float32 and np.allclose(g32, w32, rtol=1e-6, atol=10.0), np.abs(g32 - w32).max()
        # (no comparison with the fp64 vector here: the fp32 alpha the reference was given is itself 2 % off for
        #  thin layers -- 1 - x*log(r) cancels -- so the fp32 chain is only as good as its inputs)
        # a contiguous bottom-most level range (47 of 137)
        sub = F(t[90:], q[90:], al[90:], de[90:])
        assert np.allclose(sub, G["chain.f32.from_alpha_delta_n47"], rtol=1e-6, atol=1e-9)
        # device-resident, ragged column count (61): DeviceArray in -> DeviceArray out
        d = [ek.to_device(x) for x in (t, q, al.astype(npdt), de.astype(npdt))]
        dev = F(*d)
        assert isinstance(dev, ek.DeviceArray) and np.array_equal(dev.to_host(), g32, equal_nan=True)
        # vertical axis last (square-free shape): moved to the front and back
        tt = np.ascontiguousarray(np.moveaxis(t, 0, -1))
        moved = F(tt, np.moveaxis(q, 0, -1), np.moveaxis(al, 0, -1), np.moveaxis(de, 0, -1), vertical_axis=1)
        assert moved.shape == tt.shape and np.array_equal(np.moveaxis(moved, -1, 0), got)
    with pytest.raises(ValueError, match="same shape"):
        F(t, q[:5], al, de)


def test_alpha_delta_chain_on_the_device(ek):
    """pressure_on_hybrid_levels(output=alpha, delta) -> thickness_from_alpha_delta, everything resident in HBM,
    equals the fused scan on a field large enough for several workgroups (fp32: both within the reference's bar
    of the fp64 oracle)."""
    from oracle import vertical_oracle as vo

    A, B = ek.vertical.hybrid_level_parameters(137)
    rng = np.random.default_rng(8)
    npts = 5000
    sp = rng.uniform(5.2e4, 1.04e5, npts)
    pf = vo.pressure_on_hybrid_levels(A, B, sp)
    t = np.maximum(288.15 * (np.maximum(pf, 1.0) / 101325.0) ** 0.190263, 190.0) + rng.normal(0, 6, pf.shape)
    q = np.clip(rng.uniform(0, 1, pf.shape) ** 3 * 0.02 * (pf / 101325.0) ** 2, 1e-7, 0.03)
    want = vo.relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp)
    for npdt, rtol, atol in ((np.float64, 1e-6, 1e-9), (np.float32, 1e-6, 10.0)):
        a_, b_ = A.astype(npdt), B.astype(npdt)
        d_sp, d_t, d_q = (ek.to_device(x.astype(npdt)) for x in (sp, t, q))
        al, de = ek.vertical.pressure_on_hybrid_levels(a_, b_, d_sp, output=("alpha", "delta"))
        assert isinstance(al, ek.DeviceArray) and al.dtype == npdt
        z = ek.vertical.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(d_t, d_q, al, de)
        fused = ek.vertical.relative_geopotential_thickness_on_hybrid_levels(d_t, d_q, a_, b_, d_sp)
        assert isinstance(z, ek.DeviceArray) and z.shape == (137, npts)
        assert np.allclose(z.to_host(), want, rtol=rtol, atol=atol), np.abs(z.to_host() - want).max()
        assert np.allclose(fused.to_host(), want, rtol=rtol, atol=atol)


def test_example_model_level_postprocessing(ek):
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                        "model_level_postprocessing.py")
    spec = importlib.util.spec_from_file_location("example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rh, h = mod.main(nlat=37, nlon=72)
    assert rh.shape == (137, 37, 72) and np.isfinite(rh[40:]).all() and np.isfinite(h).all()
    assert (np.diff(h, axis=0) < 0).all()  # height decreases from the model top to the surface


# ---- wind.w_from_omega (the free rider on the map skeleton) ------------------------
@pytest.mark.parametrize("tag,rtol", [("f64", 1e-6), ("f32", 1e-4)])
def test_w_from_omega_vs_reference_vectors(ek, tag, rtol):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wind_golden.npz"))
    o, t, p, pl = (g[f"{tag}.in.{k}"] for k in ("omega", "t", "p", "plev"))
    W = ek.wind.array.w_from_omega
    cases = ((W(o, t, p), "field"), (W(o, t, pl[:, None]), "levmajor"), (W(o, t, o.dtype.type(85000.0)), "scalar_p"))
    for got, key in cases:
        want = g[f"{tag}.out.{key}"]
        assert got.dtype == want.dtype and got.shape == want.shape, key
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want)), key
        fin = np.isfinite(want) & (want != 0)
        assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) <= rtol, key
    # the reference's own known answer (tests/wind/test_wind.py:183-194 there), lists in, fp64 out
    got = ek.wind.w_from_omega([1.2, 21.3], [285.6, 261.1], [100000.0, 85000.0])
    assert got.dtype == np.float64 and np.allclose(got, [-0.1003208031, -1.9152219066])
    # device-resident: DeviceArray in -> DeviceArray out, level vector kept as such
    d = [ek.to_device(x) for x in (o, t, pl[:, None])]
    dev = W(*d)
    assert isinstance(dev, ek.DeviceArray) and np.array_equal(dev.to_host(), cases[1][0], equal_nan=True)


def test_geopotential_scan_in_level_chunks_equals_one_launch(ek):
    """The column scan can be cut into launches of a few levels each; the running sum crosses a chunk boundary in the
    output row above it (csrc/hybrid.hip).  Chunks of 3 and 5 levels must reproduce the single launch bit for bit."""
    from ekm_hip import _ffi

    rng = np.random.default_rng(3)
    A, B = G["coef.137.A"][137 - 24:], G["coef.137.B"][137 - 24:]
    npts = 4099  # ragged: the unaligned kernel variant too
    sp = rng.uniform(60000.0, 104000.0, npts).astype(np.float32)
    zs = rng.uniform(0.0, 3e4, npts).astype(np.float32)
    t = rng.uniform(200.0, 300.0, (24, npts)).astype(np.float32)
    q = rng.uniform(0.0, 0.02, (24, npts)).astype(np.float32)
    lib = _ffi.lib()
    try:
        outs = []
        for chunk in (1 << 20, 5, 3):
            _ffi.check(lib.ekm_set_tuning_param(b"geo_chunk_levels", chunk))
            outs.append(ek.vertical.geopotential_on_hybrid_levels(t, q, zs, A, B, sp))
            outs.append(ek.vertical.geopotential_on_hybrid_levels(t[:, :4096], q[:, :4096], zs[:4096], A, B, sp[:4096]))
    finally:
        _ffi.check(lib.ekm_set_tuning_param(b"geo_chunk_levels", 1 << 20))
    for k in (2, 4):
        assert np.array_equal(outs[0], outs[k], equal_nan=True) and np.array_equal(outs[1], outs[k + 1], equal_nan=True)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_row_ordered_and_column_kernels_agree_bit_for_bit(ek, dt):
    """pressure_on_hybrid_levels runs one workgroup per (level, tile) by default (tuning parameter hybrid_rows = 1); the
    round-2 kernel -- one lane per column, walking down the levels -- is the same arithmetic: every output, level
    selections, ragged and unaligned column counts, a zero-pressure model top."""
    from ekm_hip import _ffi

    lib = _ffi.lib()
    A, B = G["coef.137.A"].astype(dt), G["coef.137.B"].astype(dt)
    rng = np.random.default_rng(23)
    try:
        for npts, levels, output in ((4096, None, ["full", "half", "delta", "alpha"]), (1031, None, ["full"]),
                                     (8193, [1, 2, 50, 137], ["half", "alpha"]), (64 * 1024 + 2, [90, 91, 137], ["delta", "full"]),
                                     (3, None, ["full", "half"])):
            sp = ek.to_device(rng.uniform(5e4, 1.05e5, npts).astype(dt))
            got = {}
            for rows in (1, 0):
                _ffi.check(lib.ekm_set_tuning_param(b"hybrid_rows", rows))
                res = ek.vertical.pressure_on_hybrid_levels(A, B, sp, levels=levels, output=output)
                got[rows] = [r.to_host() for r in (res if isinstance(res, tuple) else (res,))]
            for name, a, b in zip(output, got[1], got[0]):
                assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), (npts, levels, name)
    finally:
        _ffi.check(lib.ekm_set_tuning_param(b"hybrid_rows", 1))


def test_special_operands_through_the_hybrid_level_functions(ek):
    """NaN, infinities, zeros, negatives, 1e-30 and 1e30 as surface pressure, as t / q at one level and as surface
    geopotential, in every combination (6561 columns x 137 levels, both dtypes), through pressure_on_hybrid_levels (every
    output, both alpha_top), the geopotential chain (thickness, geopotential, the four heights) and w_from_omega: same NaN /
    inf pattern as the oracle, finite values at the bar (fp32: max(1e-4, 4 x the reference's own fp32-vs-fp64 distance --
    its alpha = 1 - p/dp*log(..) cancels in fp32)).  tools/special_probe_vertical.py prints what differs."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "special_probe_vertical.py")
    spec = importlib.util.spec_from_file_location("special_probe_vertical", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_random_shapes_levels_outputs_and_axes_through_the_hybrid_level_functions(ek):
    """tools/shape_fuzz_vertical.py: 200 random calls -- surface-pressure shapes of 0 ... 3 dimensions, both dtypes, L91 / L137,
    level subsets, output selections, alpha_top, the vertical axis moved, Fortran-ordered views -- against the oracle, which
    agrees with the reference itself on every such call (`--reference` in the build container)."""
    import importlib.util
    import os
    import sys

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "shape_fuzz_vertical.py")
    spec = importlib.util.spec_from_file_location("shape_fuzz_vertical", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, sys.argv = sys.argv, ["shape_fuzz_vertical.py", "--trials", "200"]
    try:
        assert mod.main() == 0
    finally:
        sys.argv = argv

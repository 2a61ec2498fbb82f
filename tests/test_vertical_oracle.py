"""pressure_on_hybrid_levels: the oracle against vectors recorded from the reference (bit for bit)
and against the data of the reference's own fixture (its tolerances)."""
import json
import os

import numpy as np
import pytest

from oracle import vertical_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "vertical_golden.npz"))
CASES = json.loads(bytes(G["manifest"]).decode())


def case_args(c):
    A, B = G[f"coef.{c['nlev']}.A"].copy(), G[f"coef.{c['nlev']}.B"].copy()
    if c.get("top_offset"):
        A = A + c["top_offset"]
    dt = np.float32 if c["dtype"] == "f32" else np.float64
    sp = G[f"sp.{c['sp']}"].astype(dt)
    if c["dtype"] == "f32":
        A, B = A.astype(dt), B.astype(dt)
    return A, B, sp


@pytest.mark.parametrize("c", CASES, ids=[c["id"] for c in CASES])
def test_oracle_bit_exact(c):
    A, B, sp = case_args(c)
    res = vo.pressure_on_hybrid_levels(A, B, sp, levels=c["levels"], alpha_top=c["alpha_top"], output=c["output"],
                                       vertical_axis=c["vertical_axis"])
    res = res if isinstance(res, tuple) else (res,)
    for name, r in zip(c["output"], res):
        w = G[f"{c['id']}.{name}"]
        assert r.dtype == w.dtype and np.array_equal(r, w, equal_nan=True), (c["id"], name)


def test_oracle_vs_reference_fixture():
    """tests/vertical/test_array_vertical.py:159-213 there: atol 1e-8 / rtol 1e-6 in fp64."""
    A, B, sp = G["fixture.A"], G["fixture.B"], G["fixture.p_surf"]
    full, half, delta, alpha = vo.pressure_on_hybrid_levels(A, B, sp, output=["full", "half", "delta", "alpha"])
    for got, name in ((full, "p_full"), (half, "p_half"), (delta, "delta"), (alpha, "alpha")):
        assert np.allclose(got, G[f"fixture.{name}"], atol=1e-8, rtol=1e-6), name


def test_oracle_errors():
    A, B, sp = G["coef.137.A"], G["coef.137.B"], np.array([1e5])
    with pytest.raises(ValueError, match="Unknown output type"):
        vo.pressure_on_hybrid_levels(A, B, sp, output="bogus")
    with pytest.raises(ValueError, match="Unknown method"):
        vo.pressure_on_hybrid_levels(A, B, sp, alpha_top="bogus")
    with pytest.raises(ValueError, match="exceeds the maximum"):
        vo.pressure_on_hybrid_levels(A, B, sp, levels=[138])
    with pytest.raises(ValueError, match="starts at 1"):
        vo.pressure_on_hybrid_levels(A, B, sp, levels=[0])
    with pytest.raises(ValueError, match="At least one"):
        vo.pressure_on_hybrid_levels(A, B, sp, output=[])


CHAIN = json.loads(bytes(G["chain_manifest"]).decode())


def chain_args(c):
    dt = np.float32 if c["dtype"] == "f32" else np.float64
    A, B = G["coef.137.A"], G["coef.137.B"]
    if c["dtype"] == "f32":
        A, B = A.astype(dt), B.astype(dt)
    t, q, sp, zs = (G[f"chain.{k}"].astype(dt) for k in ("t", "q", "sp", "zs"))
    nl = c["nlev"]
    return t[137 - nl:], q[137 - nl:], zs, A, B, sp


def chain_calls(mod, t, q, zs, A, B, sp):
    out = {"thickness": mod.relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp),
           "geopotential": mod.geopotential_on_hybrid_levels(t, q, zs, A, B, sp)}
    for ht in ("geometric", "geopotential"):
        for hr in ("sea", "ground"):
            out[f"h_{ht}_{hr}"] = mod.height_on_hybrid_levels(t, q, zs, A, B, sp, h_type=ht, h_reference=hr)
    return out


@pytest.mark.parametrize("c", CHAIN, ids=[c["id"] for c in CHAIN])
def test_chain_oracle_bit_exact(c):
    for k, v in chain_calls(vo, *chain_args(c)).items():
        w = G[f"{c['id']}.{k}"]
        assert v.dtype == w.dtype and np.array_equal(v, w, equal_nan=True), (c["id"], k)


def test_chain_oracle_vs_reference_fixtures():
    """tests/vertical/test_array_vertical.py:385-523 there (fp64: atol 1e-8, rtol 1e-6)."""
    A, B, sp, t, q = (G[f"fixture.{k}"] for k in ("A", "B", "p_surf", "t", "q"))
    z = vo.relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp)
    assert np.allclose(z, G["fixture.z"], atol=1e-8, rtol=1e-6)
    assert np.allclose(vo.relative_geopotential_thickness_on_hybrid_levels(t[90:], q[90:], A, B, sp), G["fixture.z"][90:],
                       atol=1e-8, rtol=1e-6)
    al, de = vo.pressure_on_hybrid_levels(A, B, sp, output=("alpha", "delta"))
    assert np.allclose(vo.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(t, q, al, de), G["fixture.z"],
                       atol=1e-8, rtol=1e-6)
    A137, B137 = G["coef.137.A"], G["coef.137.B"]
    sp, zs, t, q = (G[f"hfix.{k}"] for k in ("p_surf", "z_surf", "t", "q"))
    for ht in ("geometric", "geopotential"):
        for hr in ("sea", "ground"):
            h = vo.height_on_hybrid_levels(t, q, zs, A137, B137, sp, h_type=ht, h_reference=hr)
            assert np.allclose(h, G[f"hfix.h_{ht}_{hr}"], atol=1e-8, rtol=1e-6), (ht, hr)


def test_from_alpha_delta_oracle_bit_exact():
    """relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta (vertical.py:815-893) on the alpha / delta
    the reference was given when the vectors were recorded: fp64; fp32 t, q with the reference's fp64 alpha / delta
    (promotes to fp64); all-fp32; a 47-level subset."""
    for dt in ("f64", "f32"):
        npdt = np.float32 if dt == "f32" else np.float64
        t, q = (G[f"chain.{k}"].astype(npdt) for k in ("t", "q"))
        al, de = G[f"chain.{dt}.alpha"], G[f"chain.{dt}.delta"]
        assert al.dtype == np.float64  # the reference hands these out in fp64 whatever the input dtype
        got = vo.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(t, q, al, de)
        w = G[f"chain.{dt}.from_alpha_delta"]
        assert got.dtype == w.dtype and np.array_equal(got, w, equal_nan=True), dt
    t, q = (G[f"chain.{k}"].astype(np.float32) for k in ("t", "q"))
    al, de = G["chain.f32.alpha"], G["chain.f32.delta"]
    got = vo.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(t, q, al.astype(np.float32), de.astype(np.float32))
    assert got.dtype == np.float32 and np.array_equal(got, G["chain.f32.from_alpha_delta_f32ad"], equal_nan=True)
    got = vo.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(t[90:], q[90:], al[90:], de[90:])
    assert np.array_equal(got, G["chain.f32.from_alpha_delta_n47"], equal_nan=True)


def test_hybrid_level_parameters_ship_inside_the_package():
    """ekm_hip.vertical.hybrid_level_parameters (vertical/array/hybrid.py:40-102): the tables the reference hands
    out, bit for bit, from a data file INSIDE the product package (no GPU, no library needed); same errors."""
    import inspect

    from ekm_hip import vertical

    for n in (137, 91):
        A, B = vertical.hybrid_level_parameters(n)
        assert A.dtype == np.float64 and A.shape == (n + 1,) and B.shape == (n + 1,)
        assert np.array_equal(A, G[f"coef.{n}.A"]) and np.array_equal(B, G[f"coef.{n}.B"])
        A2, _ = vertical.hybrid_level_parameters(str(n), model="IFS")
        assert np.array_equal(A, A2)
    with pytest.raises(ValueError, match="not available for 60 levels in model 'ifs'"):
        vertical.hybrid_level_parameters(60)
    with pytest.raises(ValueError, match="Model 'gfs' not recognized"):
        vertical.hybrid_level_parameters(137, model="gfs")
    assert str(inspect.signature(vertical.hybrid_level_parameters)) == "(n_levels, model='ifs')"
    assert str(inspect.signature(vertical.relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta)) == \
        "(t, q, alpha, delta, vertical_axis=0)"
    path = os.path.join(os.path.dirname(os.path.abspath(vertical.__file__)), "data", "ifs_levels.npz")
    assert os.path.exists(path) and "tests" not in os.path.relpath(path, os.path.dirname(os.path.dirname(vertical.__file__)))

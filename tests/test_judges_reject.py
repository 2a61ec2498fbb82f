"""Negative tests of the parity judges: every acceptance rule of tests/_compare.py and tests/_fuzz.py is fed a correct
output (must pass) and the same output with a deliberate defect (must raise AssertionError).

VERDICT r5 fed the judges wrong outputs and two of them said yes: 1 % of the bisection points moved by two quanta passed
(a flat 2 % flip budget against a measured use of 3e-8), and every result beyond 1e6 scaled by 1 + 5e-3 passed in fp32
AND fp64 (a flat 1e-2 bar for "absurd" values).  The budgets are re-based on measured use now (tests/_compare.py::
bisect_flip_allowed, tests/_fuzz.py::ABSURD_FACTOR, edge_allowed); this file keeps them able to say no.  The mutations
are the ones of that review, plus smaller ones at the edge of each budget.

Both suites run it: `twin` takes the output under test from the kernels' templates compiled for the host (no GPU),
`gpu` from the gfx950 kernels through the C ABI.
"""
import os

import numpy as np
import pytest

import _fuzz
import _hosttwin as twin
from _compare import BISECT_QUANTUM, assert_parity, bisect_flip_allowed

np.seterr(all="ignore")
N = 1 << 17
BACKENDS = [pytest.param("twin", marks=pytest.mark.skipif(not os.path.exists(twin.PATH), reason="host twin not built")),
            pytest.param("gpu", marks=pytest.mark.gpu)]
TAGS = {"f32": np.float32, "f64": np.float64}


@pytest.fixture(scope="module", params=BACKENDS)
def run(request):
    """run(func, ins, kwargs, dtype) -> the output under test, as NumPy."""
    if request.param == "twin":
        return lambda func, ins, kwargs, dtype: twin.by_reference_name(func, ins, dict(kwargs), dtype)
    ek = request.getfixturevalue("ek")
    return lambda func, ins, kwargs, dtype: getattr(ek.thermo, func)(*ins, **kwargs)


@pytest.fixture(scope="module", params=sorted(TAGS))
def points(request):
    dtype = TAGS[request.param]
    return request.param, dtype, _fuzz.make(n=N, dtype=dtype)


def _rejects(judge, got, what):
    from _compare import CENSUS, LEDGER

    marks = len(LEDGER), len(CENSUS)
    try:
        with pytest.raises(AssertionError):
            judge(got)
            pytest.fail(f"the judge ACCEPTED {what}", pytrace=False)
    finally:  # what a deliberately wrong output "used" is not part of the run's budget ledger
        del LEDGER[marks[0]:], CENSUS[marks[1]:]


def _pick(rng, mask, k):
    idx = np.flatnonzero(mask)
    assert idx.size >= k, (idx.size, k)
    return rng.choice(idx, size=k, replace=False)


# ---- the bisection (the reference's default t_method) ----------------------------------------------------------------
@pytest.mark.parametrize("method", _fuzz.METHODS)
def test_bisection_judge(run, points, method):
    tag, dtype, d = points
    func, keys = "wet_bulb_temperature_from_specific_humidity", ("t", "q", "p")
    got = np.asarray(run(func, [d[k] for k in keys], dict(ept_method=method, t_method="bisect"), dtype))
    judge = lambda g: _fuzz.judge(func, keys, method, "bisect", tag, d, g)  # noqa: E731
    judge(got.copy())  # the kernels' own output passes
    rng = np.random.default_rng(5)
    fin = np.isfinite(got)
    q = dtype(BISECT_QUANTUM)
    # VERDICT r5's mutation: 1 % of the points two quanta off (0.059 K)
    bad = got.copy()
    bad[_pick(rng, fin, N // 100)] += 2 * q
    _rejects(judge, bad, "1 % of the points moved by two quanta")
    # one more than the budget, by ONE quantum
    k = bisect_flip_allowed(tag, N) + 1 + 20  # (20: a point picked among the few reference-unstable ones does not count)
    bad = got.copy()
    bad[_pick(rng, fin, k)] -= q
    _rejects(judge, bad, f"{k} points moved by one quantum (allowed {bisect_flip_allowed(tag, N)})")
    # beyond two quanta: not a sign flip any more
    bad = got.copy()
    bad[_pick(rng, fin, 5)] += 3 * q
    _rejects(judge, bad, "5 points moved by three quanta")
    # NaN where the reference has a number, a number where it has NaN
    bad = got.copy()
    bad[_pick(rng, fin, 25)] = np.nan
    _rejects(judge, bad, "25 injected NaN")
    if (~fin).sum() >= 25:
        bad = got.copy()
        bad[_pick(rng, ~fin, 25)] = dtype(253.16)
        _rejects(judge, bad, "25 NaN replaced by a number")


def test_bisect_flip_budget_is_what_was_measured():
    """(profiles/r06_parity_budgets.txt) nothing below 100,000 points; 1e-5 (fp32) / 2e-6 (fp64) of the points above."""
    assert bisect_flip_allowed("f32", 480) == 0 and bisect_flip_allowed("f64", 99_999) == 0
    assert bisect_flip_allowed("f32", 1 << 20) == 11 and bisect_flip_allowed("f64", 1 << 20) == 3
    want = np.full(480, 253.16) + BISECT_QUANTUM * np.arange(480)
    assert_parity(want.copy(), want, "f64", "identical", bisect=True)
    bad = want.copy()
    bad[7] += BISECT_QUANTUM
    with pytest.raises(AssertionError):
        assert_parity(bad, want, "f64", "one flip in 480 rows", bisect=True)
    bad = want.copy()
    bad[7] += 0.75 * BISECT_QUANTUM  # not a lattice value at all, and above 293 K below the relative bar of fp32
    with pytest.raises(AssertionError):
        assert_parity(bad.astype(np.float32), want.astype(np.float32), "f32", "off the lattice", bisect=True)


# ---- Newton (one Davies-Jones step) -----------------------------------------------------------------------------------
@pytest.mark.parametrize("method", _fuzz.METHODS)
def test_newton_judge(run, points, method):
    tag, dtype, d = points
    func, keys = "wet_bulb_temperature_from_specific_humidity", ("t", "q", "p")
    got = np.asarray(run(func, [d[k] for k in keys], dict(ept_method=method, t_method="newton"), dtype))
    judge = lambda g: _fuzz.judge(func, keys, method, "newton", tag, d, g)  # noqa: E731
    judge(got.copy())
    rng = np.random.default_rng(6)
    # physical points: an atmospheric result reached by a correction of a few kelvin (the reference's own inputs tell)
    from oracle import thermo_oracle as orc

    ins64 = [d[k].astype(np.float64) for k in keys]
    w64 = orc.wet_bulb_temperature_from_specific_humidity(*ins64, ept_method=method, t_method="newton")
    phys = np.isfinite(got) & (w64 > 200.0) & (w64 < 330.0) & (np.abs(w64 - ins64[0]) < 30.0) & (ins64[2] > 3e4) & (ins64[1] < 0.03)
    bad = got.copy()
    bad[_pick(rng, phys, 5)] *= dtype(1 + 1e-3)
    _rejects(judge, bad, "5 physical points scaled by 1 + 1e-3")
    bad = got.copy()
    bad[_pick(rng, phys, 1)] = np.nan
    _rejects(judge, bad, "one injected NaN on a physical point")
    if tag == "f64":
        bad = got * (1 + 5e-7)
        _rejects(judge, bad, "every fp64 point scaled by 1 + 5e-7")
    else:
        bad = got * dtype(1 + 3e-4)
        _rejects(judge, bad, "every fp32 point scaled by 1 + 3e-4")


# ---- the closed-form functions, with the results beyond any thermodynamic quantity ----------------------------------
DIRECT = [("relative_humidity_from_specific_humidity", ("t", "q", "p"), {}),
          ("saturation_ept", ("t", "p"), {"method": "bolton35"}),
          ("ept_from_specific_humidity", ("t", "q", "p"), {"method": "bolton39"}),
          ("wet_bulb_potential_temperature_from_specific_humidity", ("t", "q", "p"), {"ept_method": "ifs", "t_method": "direct"})]


@pytest.mark.parametrize("func,keys,kwargs", DIRECT, ids=[f for f, _, _ in DIRECT])
def test_direct_judge(run, points, func, keys, kwargs):
    tag, dtype, d = points
    got = np.asarray(run(func, [d[k] for k in keys], kwargs, dtype))
    judge = lambda g: _fuzz.judge_direct(func, keys, kwargs, tag, d, g)  # noqa: E731
    judge(got.copy())
    rng = np.random.default_rng(7)
    absurd = np.isfinite(got) & (np.abs(got.astype(np.float64)) > _fuzz.ABSURD)
    assert absurd.sum() > 1000, "this case is here for its results beyond 1e6"
    # VERDICT r5's mutation: every result beyond 1e6 scaled by 1 + 5e-3 (round 5 accepted it in both dtypes)
    bad = got.copy()
    bad[absurd] *= got.dtype.type(1 + 5e-3)
    _rejects(judge, bad, "every result beyond 1e6 scaled by 1 + 5e-3")
    if tag == "f64":
        bad = got.copy()
        bad[absurd] *= 1 + 5e-7
        _rejects(judge, bad, "every fp64 result beyond 1e6 scaled by 1 + 5e-7")
    # the ordinary results: beyond the plain bar on 1 % of the points
    ordinary = np.isfinite(got) & ~absurd & (got != 0)
    bad = got.copy()
    bad[_pick(rng, ordinary, N // 100)] *= got.dtype.type(1 + (5e-7 if tag == "f64" else 3e-4))
    _rejects(judge, bad, "1 % of the ordinary results beyond the plain bar")
    bad = got.copy()
    bad[_pick(rng, ordinary, 1)] = np.nan
    _rejects(judge, bad, "one injected NaN")


def test_fused_judge(run, points):
    """BASELINE config 5's six outputs: each is held to the separate function's judge."""
    tag, dtype, d = points
    outs = [np.asarray(o) for o in run("pipeline_full", [d[k] for k in ("t", "q", "p")], {}, dtype)]
    _fuzz.judge_fused("pipeline_full", tag, d, [o.copy() for o in outs])
    rng = np.random.default_rng(8)
    for k in range(6):
        ok = np.isfinite(outs[k]) & (outs[k] != 0) & (np.abs(outs[k].astype(np.float64)) < _fuzz.ABSURD)
        if k == 5:  # tw: where the reference's one Newton step is well conditioned
            ok &= (d["p"] > 3e4) & (d["q"] < 0.03) & (outs[k] > 200) & (outs[k] < 330)
        bad = [o.copy() for o in outs]
        bad[k][_pick(rng, ok, 200)] *= outs[k].dtype.type(1 + (5e-6 if tag == "f64" else 1e-3))
        _rejects(lambda g: _fuzz.judge_fused("pipeline_full", tag, d, g), bad, f"output {k}: 200 points off")


# ---- the plain comparison the golden tests use --------------------------------------------------------------------
@pytest.mark.parametrize("tag", sorted(TAGS))
def test_plain_assert_parity(tag):
    dtype = TAGS[tag]
    want = np.linspace(200.0, 320.0, 4096).astype(dtype)
    assert_parity(want.copy(), want, tag, "identical")
    eps = 2e-6 if tag == "f64" else 2e-4  # just beyond the north-star bars (1e-6 / 1e-4)
    bad = want.copy()
    bad[17] *= dtype(1 + eps)
    with pytest.raises(AssertionError):
        assert_parity(bad, want, tag, "one point beyond the bar")
    bad = want.copy()
    bad[17] = np.nan
    with pytest.raises(AssertionError):
        assert_parity(bad, want, tag, "one NaN")
    bad = want.copy()
    bad[17] = np.inf
    with pytest.raises(AssertionError):
        assert_parity(bad, want, tag, "one inf")
    # the ill-conditioned allowance (max(rtol, 4*delta) where the reference's fp32 and fp64 runs differ) is counted
    if tag == "f32":
        ref64 = want.astype(np.float64)
        ref64[:40] *= 1 + 1e-3  # the reference disagrees with itself on 40 of 4096 points: more than the 1e-4 of the points allowed
        with pytest.raises(AssertionError):
            assert_parity(want.copy(), want, tag, "too many ill-conditioned points", ref64=ref64)


def test_special_operand_judge(run):
    func, keys, kwargs = "potential_temperature", ("t", "p"), {}
    for tag, dtype in TAGS.items():
        ins = _fuzz.special_operands(keys, dtype)
        got = np.asarray(run(func, ins, kwargs, dtype))
        _fuzz.judge_special(func, keys, kwargs, tag, ins, got.copy())
        fin = np.flatnonzero(np.isfinite(got) & (got != 0))
        bad = got.copy()
        bad[fin[0]] *= dtype(1 + 1e-3)
        _rejects(lambda g: _fuzz.judge_special(func, keys, kwargs, tag, ins, g), bad, "a finite result off by 1e-3")
        bad = got.copy()
        bad[np.flatnonzero(np.isnan(got))[0]] = dtype(1.0)
        _rejects(lambda g: _fuzz.judge_special(func, keys, kwargs, tag, ins, g), bad, "a number where the reference has NaN")
        inf = np.flatnonzero(np.isinf(got))
        if inf.size:
            bad = got.copy()
            bad[inf[0]] = -bad[inf[0]]
            _rejects(lambda g: _fuzz.judge_special(func, keys, kwargs, tag, ins, g), bad, "an infinity with the wrong sign")

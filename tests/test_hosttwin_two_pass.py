"""fp64 two-pass scheme of the kernels (csrc/map_kernel.hpp::apply_points, csrc/thermo_math.hpp::fdouble): a first pass
whose primitives carry no IEEE special-operand fix-ups and poison to NaN instead, then plain double for the points with a
non-finite output.  The claim it rests on -- a point whose first-pass outputs are all finite got exactly the plain-double
result -- is checked here on the host twin, for every function and variant, on operands chosen to hit the fix-ups:
zeros, infinities, NaN, negative and denormal values, the thresholds of the formulas, huge and tiny magnitudes, mixed into
ordinary atmospheric values so that each point has ONE odd operand at a time and then several."""
import os
import sys

import numpy as np
import pytest

import _hosttwin

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from _case_table import case_table  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(_hosttwin.PATH), reason="host twin not built (run make)")

ODD = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, -250.0, 5e-324, 2.3e-308, 1e-300, 1e300, 1.7e308, -1e300,
                32.19, -0.7, 273.16, 250.16, 56.0, 1e5, 1e-4, 0.621981, 1.0, 0.4, 1e38, 3.5e38, 1e-38, 1e-46])


def _typical(name, n, rng):
    if name in ("t", "t2", "td", "th", "ept"):
        return rng.uniform(200.0, 320.0, n)
    if name == "tc":
        return rng.uniform(-70.0, 45.0, n)
    if name in ("q", "w"):
        return 10.0 ** rng.uniform(-6.0, -1.6, n)
    if name in ("p", "p2"):
        return 10.0 ** rng.uniform(2.0, 5.03, n)
    if name in ("e", "es"):
        return 10.0 ** rng.uniform(-1.0, 3.8, n)
    if name == "r":
        return rng.uniform(0.0, 110.0, n)
    raise KeyError(name)


CASES = case_table()


IDS = [f"{f}.{'.'.join(map(str, k.values()))}" for f, _, k in CASES]


def adversarial_args(func, argnames):
    """Operand arrays for one function: each argument in turn swept over ODD (the others ordinary), then several odd
    operands per point, then none."""
    import zlib

    rng = np.random.default_rng(zlib.crc32(func.encode()))
    nodd, narg = ODD.size, len(argnames)
    blocks = []
    base = 64
    for a in range(narg):  # one odd operand at a time
        cols = [_typical(nm, nodd * base, rng) for nm in argnames]
        cols[a] = np.repeat(ODD, base)
        blocks.append(cols)
    # NaN in each operand in turn BESIDE an IEEE special (or any other odd value) in each of the others (the rule
    # that a NaN input explains a non-finite fast-pass output, ops.hpp::two_pass_redo_needed / OpDeps, must not leave the fast
    # pass's NaN standing where the formula ignores that operand and the plain pass returns a number or an infinity)
    if narg >= 2:
        import itertools

        for a in range(narg):
            others = [i for i in range(narg) if i != a]
            combos = np.array(list(itertools.product(ODD, repeat=len(others))) if len(others) <= 2 else
                              [c for c in itertools.product(ODD[[0, 2, 3, 4, 5, 10, 20]], repeat=len(others))])
            cols = [None] * narg
            cols[a] = np.full(len(combos), np.nan)
            for j, i in enumerate(others):
                cols[i] = combos[:, j].copy()
            blocks.append(cols)
            for i in others:  # ... and beside ONE special, the rest ordinary
                cols = [_typical(nm, nodd, rng) for nm in argnames]
                cols[a] = np.full(nodd, np.nan)
                cols[i] = ODD.copy()
                blocks.append(cols)
    cols = [_typical(nm, 4096, rng) for nm in argnames]  # several at once
    for c in cols:
        idx = rng.random(c.size) < 0.4
        c[idx] = rng.choice(ODD, int(idx.sum()))
    blocks.append(cols)
    blocks.append([_typical(nm, 4096, rng) for nm in argnames])  # none
    return [np.concatenate([b[a] for b in blocks]) for a in range(narg)]


def assert_bit_equal(func, kwargs, args, plain, two):
    plain = plain if isinstance(plain, tuple) else (plain,)
    two = two if isinstance(two, tuple) else (two,)
    for k, (a, b) in enumerate(zip(plain, two)):
        same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        bad = np.flatnonzero(~same)
        assert bad.size == 0, (f"{func} {kwargs} out{k}: {bad.size} points differ, first at {bad[0]}: "
                               f"inputs {[x[bad[0]] for x in args]} plain {a[bad[0]]!r} two-pass {b[bad[0]]!r}")


@pytest.mark.parametrize("func,argnames,kwargs", CASES, ids=IDS)
def test_two_pass_equals_plain_double(func, argnames, kwargs, monkeypatch):
    args = adversarial_args(func, argnames)
    with np.errstate(all="ignore"):
        monkeypatch.setenv("EKM_TWIN_PLAIN_F64", "1")
        plain = _hosttwin.by_reference_name(func, args, kwargs, np.float64)
        monkeypatch.setenv("EKM_TWIN_PLAIN_F64", "0")
        two = _hosttwin.by_reference_name(func, args, kwargs, np.float64)
    assert_bit_equal(func, kwargs, args, plain, two)

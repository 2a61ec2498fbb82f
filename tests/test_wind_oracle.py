"""`wind.w_from_omega` (free rider): oracle and host twin against the vectors
recorded from the reference (tests/golden/gen_golden_wind.py) and the reference's own known answer."""
import os

import numpy as np
import pytest

import _hosttwin
from oracle import wind_oracle as wo

np.seterr(all="ignore")
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wind_golden.npz"))


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_reproduces_the_reference_bit_for_bit(tag):
    o, t, p, pl = (G[f"{tag}.in.{k}"] for k in ("omega", "t", "p", "plev"))
    for got, key in ((wo.w_from_omega(o, t, p), "field"), (wo.w_from_omega(o, t, pl[:, None]), "levmajor"),
                     (wo.w_from_omega(o, t, o.dtype.type(85000.0)), "scalar_p")):
        want = G[f"{tag}.out.{key}"]
        assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), key


def test_reference_known_answer():
    got = wo.w_from_omega(G["kat.omega"], G["kat.t"], G["kat.p"])
    assert np.allclose(got, G["kat.expected"]) and np.array_equal(got, G["kat.out"])  # tests/wind/test_wind.py:183-194


@pytest.mark.skipif(not os.path.exists(_hosttwin.PATH), reason="host twin not built (run make)")
@pytest.mark.parametrize("tag,dt,rtol", [("f64", np.float64, 1e-6), ("f32", np.float32, 1e-4)])
def test_kernel_math_vs_reference(tag, dt, rtol):
    o, t, p = (G[f"{tag}.in.{k}"] for k in ("omega", "t", "p"))
    got = _hosttwin.call("w_from_omega", (o, t, p), dtype=dt)[0]
    want = G[f"{tag}.out.field"]
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want) & (want != 0)
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) <= rtol
    assert np.array_equal(got[want == 0], want[want == 0])

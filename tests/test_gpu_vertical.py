"""pressure_on_hybrid_levels on the GPU (ekm_hip.vertical) against the recorded reference vectors.
fp64: 1e-6 relative (measured ~1e-15).  fp32: the reference's own fp32 tolerances for this function
(tests/vertical/test_array_vertical.py:192-198 there): p 1e-4 relative, delta atol 1e-6 rtol 1e-5,
alpha atol 1e-4 rtol 1e-5 (alpha = 1 - x*delta cancels to ~1 % of its terms)."""
import numpy as np
import pytest

from test_vertical_oracle import CASES, G, case_args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ek():
    import ekm_hip
    import ekm_hip.vertical  # noqa: F401

    assert ekm_hip.device_count() >= 1
    return ekm_hip


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

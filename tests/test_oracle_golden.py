"""The CPU oracle is pinned to the reference: bit-for-bit on the vectors recorded
from the reference itself, and at the reference's own tolerances on the data of
its CSV fixtures and its inline known-answer vectors."""
import numpy as np
import pytest

from _compare import assert_parity
from _golden import case_inputs, case_outputs, manifest, ref_csv
from golden.known_answers import CASES as KAT
from oracle import thermo_oracle as orc

np.seterr(all="ignore")
CASES = manifest()


def test_manifest_covers_every_public_function():
    assert {c["func"] for c in CASES} == set(orc.ALL_FUNCTIONS)
    assert len(orc.ALL_FUNCTIONS) == 39


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_oracle_bit_exact_vs_reference(case):
    out = getattr(orc, case["func"])(*[a.copy() for a in case_inputs(case)], **case["kwargs"])
    outs = out if isinstance(out, tuple) else (out,)
    for o, g in zip(outs, case_outputs(case)):
        o = np.asarray(o)
        assert o.dtype == g.dtype
        assert np.array_equal(o, g, equal_nan=True), case["id"]


@pytest.mark.parametrize("func,args,kwargs,expect,rtol", KAT, ids=[f"{i}-{c[0]}" for i, c in enumerate(KAT)])
def test_oracle_known_answers(func, args, kwargs, expect, rtol):
    out = getattr(orc, func)(*[np.asarray(a, dtype=np.float64) for a in args], **kwargs)
    outs = out if isinstance(out, tuple) else (out,)
    exps = expect if isinstance(expect, tuple) else (expect,)
    for o, e in zip(outs, exps):
        assert np.allclose(o, np.asarray(e, dtype=np.float64), rtol=max(rtol, 1e-7), atol=1e-8, equal_nan=True)


def test_oracle_vs_reference_csv_fixtures():
    """Same checks the reference's tests make on its tests/data/*.csv (default allclose;
    rtol=1e-3 for the wet-bulb files, tests/thermo/test_thermo.py:802-849 there)."""
    c = ref_csv()
    t, td, q, p = (c[f"t_hum_p_data.{k}"] for k in ("t", "td", "q", "p"))
    for ph in ("mixed", "water", "ice"):
        assert np.allclose(orc.saturation_vapour_pressure(c["sat_vp.t"], phase=ph), c[f"sat_vp.{ph}"])
        assert np.allclose(orc.saturation_vapour_pressure_slope(c["sat_vp_slope.t"], phase=ph), c[f"sat_vp_slope.{ph}"])
        assert np.allclose(orc.saturation_mixing_ratio(c["sat_mr.t"], c["sat_mr.p"], phase=ph), c[f"sat_mr.{ph}"])
        assert np.allclose(orc.saturation_specific_humidity(c["sat_q.t"], c["sat_q.p"], phase=ph), c[f"sat_q.{ph}"])
        assert np.allclose(orc.saturation_mixing_ratio_slope(c["sat_mr_slope.t"], c["sat_mr_slope.p"], phase=ph),
                           c[f"sat_mr_slope.{ph}"])
        assert np.allclose(orc.saturation_specific_humidity_slope(c["sat_q_slope.t"], c["sat_q_slope.p"], phase=ph),
                           c[f"sat_q_slope.{ph}"])
    for m in ("ifs", "bolton35", "bolton39"):
        assert np.allclose(orc.ept_from_dewpoint(t, td, p, method=m), c[f"eqpt.{m}_td"])
        assert np.allclose(orc.ept_from_specific_humidity(t, q, p, method=m), c[f"eqpt.{m}_q"])
        assert np.allclose(orc.saturation_ept(t, p, method=m), c[f"seqpt.{m}"])
        for tm in ("bisect", "newton"):
            got = orc.temperature_on_moist_adiabat(c["t_on_most_adiabat.ept"], c["t_on_most_adiabat.p"], ept_method=m, t_method=tm)
            assert np.allclose(got, c[f"t_on_most_adiabat.{m}_{tm}"], equal_nan=True)
            got = orc.wet_bulb_temperature_from_dewpoint(t, td, p, ept_method=m, t_method=tm)
            assert np.allclose(got, c[f"t_wet.{m}_{tm}_td"], rtol=1e-3, atol=0, equal_nan=True)
            got = orc.wet_bulb_temperature_from_specific_humidity(t, q, p, ept_method=m, t_method=tm)
            assert np.allclose(got, c[f"t_wet.{m}_{tm}_q"], rtol=1e-3, atol=0, equal_nan=True)
        for tm in ("direct", "bisect", "newton"):
            got = orc.wet_bulb_potential_temperature_from_dewpoint(t, td, p, ept_method=m, t_method=tm)
            assert np.allclose(got, c[f"t_wetpt.{m}_{tm}_td"], rtol=1e-3, atol=0, equal_nan=True)
            got = orc.wet_bulb_potential_temperature_from_specific_humidity(t, q, p, ept_method=m, t_method=tm)
            assert np.allclose(got, c[f"t_wetpt.{m}_{tm}_q"], rtol=1e-3, atol=0, equal_nan=True)


def test_oracle_error_conventions():
    t = np.array([280.0])
    assert orc.saturation_vapour_pressure(t, phase="bogus") is None
    with pytest.raises(KeyError):
        orc.ept_from_dewpoint(t, t - 2, np.array([9e4]), method="bogus")
    with pytest.raises(ValueError):
        orc.temperature_on_moist_adiabat(t, np.array([9e4]), t_method="bogus")
    with pytest.raises(ValueError):
        orc.lcl_temperature(t, t - 2, method="bogus")
    with pytest.raises(ValueError):
        orc.specific_humidity_from_vapour_pressure(t, t, eps=0)

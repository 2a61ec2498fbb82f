"""`earthkit.meteo.thermo` on MI355X: the reference's 39 array-level signatures
(names, argument order, keyword names and defaults of
/root/reference/src/earthkit/meteo/thermo/array/thermo.py, re-exported by
thermo/thermo.py:13-166), each backed by one fused HIP kernel launch through
libekm_thermo.so.  Inputs may be NumPy arrays / scalars / lists (computed on
the GPU, NumPy returned) or `ekm_hip.DeviceArray`s (result stays on the GPU).

Error behaviour follows the reference: `ValueError` for eps <= 0 and for an
unknown `t_method` / LCL `method`, `KeyError` for an unknown ept method, `None`
for an unknown `phase` (es_comp.py:74-79); numerical failure is NaN in-band.
There is no CPU path: without the library or a GPU every function raises.
"""
from . import _engine
from ._ffi import EPT_METHOD, LCL_METHOD, PHASE, T_METHOD

__all__ = [
    "celsius_to_kelvin", "kelvin_to_celsius", "specific_humidity_from_mixing_ratio",
    "mixing_ratio_from_specific_humidity", "vapour_pressure_from_specific_humidity",
    "vapour_pressure_from_mixing_ratio", "specific_humidity_from_vapour_pressure",
    "mixing_ratio_from_vapour_pressure", "saturation_vapour_pressure", "saturation_mixing_ratio",
    "saturation_specific_humidity", "saturation_vapour_pressure_slope", "saturation_mixing_ratio_slope",
    "saturation_specific_humidity_slope", "temperature_from_saturation_vapour_pressure",
    "relative_humidity_from_dewpoint", "relative_humidity_from_specific_humidity",
    "specific_humidity_from_dewpoint", "mixing_ratio_from_dewpoint",
    "specific_humidity_from_relative_humidity", "dewpoint_from_relative_humidity",
    "dewpoint_from_specific_humidity", "virtual_temperature", "virtual_potential_temperature",
    "potential_temperature", "temperature_from_potential_temperature", "pressure_on_dry_adiabat",
    "temperature_on_dry_adiabat", "lcl_temperature", "lcl", "ept_from_dewpoint",
    "ept_from_specific_humidity", "saturation_ept", "temperature_on_moist_adiabat",
    "wet_bulb_temperature_from_dewpoint", "wet_bulb_temperature_from_specific_humidity",
    "wet_bulb_potential_temperature_from_dewpoint", "wet_bulb_potential_temperature_from_specific_humidity",
    "specific_gas_constant", "pipeline_svp_td_rh", "pipeline_full",
]


def _one(name, args, ints=(), eps=None):
    return _engine.run(name, args, ints, eps)[0]


def _check_eps(fname, eps):
    if eps <= 0:
        raise ValueError(f"{fname}(): eps={eps} must be > 0")


def _ept_enum(method):
    return EPT_METHOD[method]  # KeyError(method) like `_EptComp.CM[method]` (thermo.py:1026)


def _t_enum(t_method, allow_direct=False):
    if t_method in ("bisect", "newton") or (allow_direct and t_method == "direct"):
        return T_METHOD[t_method]
    raise ValueError(f"temperature_on_moist_adiabat: invalid t_method={t_method} specified!")


def celsius_to_kelvin(t):
    """thermo.py:21-35"""
    return _one("celsius_to_kelvin", (t,))


def kelvin_to_celsius(t):
    """thermo.py:38-52"""
    return _one("kelvin_to_celsius", (t,))


def specific_humidity_from_mixing_ratio(w):
    """thermo.py:55-77"""
    return _one("specific_humidity_from_mixing_ratio", (w,))


def mixing_ratio_from_specific_humidity(q):
    """thermo.py:80-102"""
    return _one("mixing_ratio_from_specific_humidity", (q,))


def vapour_pressure_from_specific_humidity(q, p):
    """thermo.py:105-131"""
    return _one("vapour_pressure_from_specific_humidity", (q, p))


def vapour_pressure_from_mixing_ratio(w, p):
    """thermo.py:134-159"""
    return _one("vapour_pressure_from_mixing_ratio", (w, p))


def specific_humidity_from_vapour_pressure(e, p, eps=1e-4):
    """thermo.py:162-196"""
    _check_eps("specific_humidity_from_vapour_pressure", eps)
    return _one("specific_humidity_from_vapour_pressure", (e, p), eps=eps)


def mixing_ratio_from_vapour_pressure(e, p, eps=1e-4):
    """thermo.py:199-232"""
    _check_eps("mixing_ratio_from_vapour_pressure", eps)
    return _one("mixing_ratio_from_vapour_pressure", (e, p), eps=eps)


def saturation_vapour_pressure(t, phase="mixed"):
    """thermo.py:235-279 (unknown phase returns None, es_comp.py:74-79)"""
    if phase not in PHASE:
        return None
    return _one("saturation_vapour_pressure", (t,), (PHASE[phase],))


def saturation_mixing_ratio(t, p, phase="mixed"):
    """thermo.py:282-310"""
    if phase not in PHASE:
        raise TypeError(f"saturation_mixing_ratio(): invalid phase={phase}")  # reference fails on None arithmetic
    return _one("saturation_mixing_ratio", (t, p), (PHASE[phase],))


def saturation_specific_humidity(t, p, phase="mixed"):
    """thermo.py:313-341"""
    if phase not in PHASE:
        raise TypeError(f"saturation_specific_humidity(): invalid phase={phase}")
    return _one("saturation_specific_humidity", (t, p), (PHASE[phase],))


def saturation_vapour_pressure_slope(t, phase="mixed"):
    """thermo.py:344-364"""
    if phase not in PHASE:
        return None
    return _one("saturation_vapour_pressure_slope", (t,), (PHASE[phase],))


def _slope(kind, t, p, es, es_slope, phase, eps):
    fname = f"saturation_{kind}_slope"
    _check_eps(fname, eps)
    if es is None and es_slope is None:
        if phase not in PHASE:
            raise TypeError(f"{fname}(): invalid phase={phase}")
        return _one(fname, (t, p), (PHASE[phase],), eps=eps)
    if es is None:
        es = saturation_vapour_pressure(t, phase=phase)
    if es_slope is None:
        es_slope = saturation_vapour_pressure_slope(t, phase=phase)
    return _one(fname + "_from_es", (p, es, es_slope), eps=eps)


def saturation_mixing_ratio_slope(t, p, es=None, es_slope=None, phase="mixed", eps=1e-4):
    """thermo.py:367-415"""
    return _slope("mixing_ratio", t, p, es, es_slope, phase, eps)


def saturation_specific_humidity_slope(t, p, es=None, es_slope=None, phase="mixed", eps=1e-4):
    """thermo.py:418-467"""
    return _slope("specific_humidity", t, p, es, es_slope, phase, eps)


def temperature_from_saturation_vapour_pressure(es):
    """thermo.py:470-491"""
    return _one("temperature_from_saturation_vapour_pressure", (es,))


def relative_humidity_from_dewpoint(t, td):
    """thermo.py:494-521"""
    return _one("relative_humidity_from_dewpoint", (t, td))


def relative_humidity_from_specific_humidity(t, q, p):
    """thermo.py:524-556"""
    return _one("relative_humidity_from_specific_humidity", (t, q, p))


def specific_humidity_from_dewpoint(td, p):
    """thermo.py:559-591"""
    return _one("specific_humidity_from_dewpoint", (td, p))


def mixing_ratio_from_dewpoint(td, p):
    """thermo.py:594-626"""
    return _one("mixing_ratio_from_dewpoint", (td, p))


def specific_humidity_from_relative_humidity(t, r, p):
    """thermo.py:629-663"""
    return _one("specific_humidity_from_relative_humidity", (t, r, p))


def dewpoint_from_relative_humidity(t, r):
    """thermo.py:666-699"""
    return _one("dewpoint_from_relative_humidity", (t, r))


def dewpoint_from_specific_humidity(q, p):
    """thermo.py:702-735"""
    return _one("dewpoint_from_specific_humidity", (q, p))


def virtual_temperature(t, q):
    """thermo.py:738-764"""
    return _one("virtual_temperature", (t, q))


def virtual_potential_temperature(t, q, p):
    """thermo.py:767-798"""
    return _one("virtual_potential_temperature", (t, q, p))


def potential_temperature(t, p):
    """thermo.py:801-829"""
    return _one("potential_temperature", (t, p))


def temperature_from_potential_temperature(th, p):
    """thermo.py:832-858"""
    return _one("temperature_from_potential_temperature", (th, p))


def pressure_on_dry_adiabat(t, t_def, p_def):
    """thermo.py:861-889"""
    return _one("pressure_on_dry_adiabat", (t, t_def, p_def))


def temperature_on_dry_adiabat(p, t_def, p_def):
    """thermo.py:892-920"""
    return _one("temperature_on_dry_adiabat", (p, t_def, p_def))


def _lcl_enum(method):
    if method not in LCL_METHOD:
        raise ValueError(f"lcl_temperature: invalid method={method} specified!")
    return LCL_METHOD[method]


def lcl_temperature(t, td, method="davies"):
    """thermo.py:923-968"""
    return _one("lcl_temperature", (t, td), (_lcl_enum(method),))


def lcl(t, td, p, method="davies"):
    """thermo.py:971-1000; returns (t_lcl, p_lcl)"""
    return _engine.run("lcl", (t, td, p), (_lcl_enum(method),))


def ept_from_dewpoint(t, td, p, method="ifs"):
    """thermo.py:1326-1387"""
    return _one("ept_from_dewpoint", (t, td, p), (_ept_enum(method),))


def ept_from_specific_humidity(t, q, p, method="ifs"):
    """thermo.py:1390-1415"""
    return _one("ept_from_specific_humidity", (t, q, p), (_ept_enum(method),))


def saturation_ept(t, p, method="ifs"):
    """thermo.py:1418-1469"""
    return _one("saturation_ept", (t, p), (_ept_enum(method),))


def temperature_on_moist_adiabat(ept, p, ept_method="ifs", t_method="bisect"):
    """thermo.py:1472-1509 (accepts N-d and scalar inputs, a superset of the reference)"""
    m = _ept_enum(ept_method)
    return _one("temperature_on_moist_adiabat", (ept, p), (m, _t_enum(t_method)))


def wet_bulb_temperature_from_dewpoint(t, td, p, ept_method="ifs", t_method="bisect"):
    """thermo.py:1512-1549"""
    m = _ept_enum(ept_method)
    return _one("wet_bulb_temperature_from_dewpoint", (t, td, p), (m, _t_enum(t_method)))


def wet_bulb_temperature_from_specific_humidity(t, q, p, ept_method="ifs", t_method="bisect"):
    """thermo.py:1552-1590"""
    m = _ept_enum(ept_method)
    return _one("wet_bulb_temperature_from_specific_humidity", (t, q, p), (m, _t_enum(t_method)))


def wet_bulb_potential_temperature_from_dewpoint(t, td, p, ept_method="ifs", t_method="direct"):
    """thermo.py:1593-1634"""
    m = _ept_enum(ept_method)
    return _one("wet_bulb_potential_temperature_from_dewpoint", (t, td, p), (m, _t_enum(t_method, True)))


def wet_bulb_potential_temperature_from_specific_humidity(t, q, p, ept_method="ifs", t_method="direct"):
    """thermo.py:1637-1675"""
    m = _ept_enum(ept_method)
    return _one("wet_bulb_potential_temperature_from_specific_humidity", (t, q, p), (m, _t_enum(t_method, True)))


def specific_gas_constant(q):
    """thermo.py:1678-1707"""
    return _one("specific_gas_constant", (q,))


# ---- fused compositions (one read of t, q, p; one write per output) ---------
def pipeline_svp_td_rh(t, q, p):
    """(es, td, rh) = saturation_vapour_pressure(t), dewpoint_from_specific_humidity(q, p),
    relative_humidity_from_specific_humidity(t, q, p) in a single pass."""
    return _engine.run("pipeline_svp_td_rh", (t, q, p))


def pipeline_full(t, q, p):
    """(theta, es, rh, td, theta_e, tw): potential_temperature, saturation_vapour_pressure,
    relative_humidity_from_specific_humidity, dewpoint_from_specific_humidity,
    ept_from_specific_humidity(method="ifs") and
    wet_bulb_temperature_from_specific_humidity(ept_method="ifs", t_method="newton") in a single pass."""
    return _engine.run("pipeline_full", (t, q, p))


# `earthkit.meteo.thermo.array.<name>` is how the reference (and its tests) reach the array-level
# functions (thermo/__init__.py:19, thermo/array/__init__.py:14): same module here.
import sys as _sys  # noqa: E402

array = _sys.modules[__name__]

"""`earthkit.meteo.wind.w_from_omega` on MI355X -- the one wind function SURVEY.md section 8f names as a free
rider on the map-kernel skeleton (a three-input elementwise map; with `p` a level vector it runs through the
same per-level kernel as the thermo functions).  Same signature as the reference
(/root/reference/src/earthkit/meteo/wind/array/wind.py:192-222); NumPy in -> NumPy out, DeviceArray in ->
DeviceArray out.  The rest of `wind` (speed, direction, polar/xy conversions, coriolis, windrose) is outside
the hot path and not built."""
import sys as _sys

from . import _engine


def w_from_omega(omega, t, p):
    """Hydrostatic vertical velocity (m/s) from pressure velocity omega (Pa/s), temperature t (K) and
    pressure p (Pa): w = -(omega * t * Rd) / (p * g), evaluated as (-Rd/g) * (omega * t / p) (wind.py:222)."""
    return _engine.run("w_from_omega", (omega, t, p))[0]


# `earthkit.meteo.wind.array.<name>` is how the reference reaches the array-level functions
array = _sys.modules[__name__]

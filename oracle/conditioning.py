"""Where the reference's own algorithm takes rounding-determined decisions -- TEST INFRASTRUCTURE.

Two selectors of the moist-adiabat inversion are discontinuous in their inputs:

* bisection (thermo.py:1055-1079): ``sign(residual)`` at each of 12 lattice points;
* Davies-Jones Newton (thermo.py:1114-1128): the initial-guess regime chosen by comparing
  ``c_te`` with ``D(p)``, 1 and 0.4.  At 10 hPa the regime-1 and regime-2 guesses differ by ~2 K and
  one Newton step leaves ~0.17 K (7e-4) of that, so a point whose ``c_te`` is within rounding of a
  threshold legitimately lands on either side in any fp32 implementation, the reference's included.

The masks below identify such points from the oracle evaluated in fp64 -- never from the output
under test -- so that parity checks can exclude and count them.
"""
import numpy as np

from . import thermo_oracle as orc


def _ept_and_p(func, inputs, kwargs):
    m = kwargs.get("ept_method", "ifs")
    a = [np.asarray(x, dtype=np.float64) for x in inputs]
    if func in ("temperature_on_moist_adiabat",):
        ept, p = a
    elif func in ("pipeline_full",):
        ept, p = orc.ept_from_specific_humidity(a[0], a[1], a[2], method="ifs"), a[2]
    else:
        f = orc.ept_from_dewpoint if "dewpoint" in func else orc.ept_from_specific_humidity
        ept = f(a[0], a[1], a[2], method=m)
        p = a[2] if "potential" not in func else np.full_like(ept, orc.p0)
    ept, p = np.broadcast_arrays(ept, p)
    return ept.ravel().copy(), p.ravel().copy(), m


def bisect_sign_noise(func, inputs, kwargs, thresh):
    """Points whose bisection meets a residual |ept*exp(-G) - th_sat| <= thresh*th_sat on the
    oracle's own fp64 path (e.g. exactly saturated input, where the residual at the second lattice
    point is mathematically zero): ``sign()`` is noise there and one early flip can end in the
    ``p - es < eps`` NaN region."""
    with np.errstate(all="ignore"):
        ept, p, m = _ept_and_p(func, inputs, kwargs)
        meth = orc._EPT[m]
        t = np.full(ept.size, orc.T0 - 20.0)
        dt = 120.0
        noisy = np.zeros(ept.size, dtype=bool)
        for _ in range(12):
            st = orc._state(t=t, p=p)
            dt /= 2.0
            g = meth["gsat"](st, scale=-1.0)
            th = meth["thsat"](st)
            r = ept * np.exp(g) - th
            noisy |= np.abs(r) <= thresh * np.abs(th)
            t = t + np.sign(r) * dt
    return noisy


def newton_regime_boundary(func, inputs, kwargs, thresh):
    """Points whose c_te lies within relative `thresh` of a regime threshold (D(p), 1, 0.4)."""
    with np.errstate(all="ignore"):
        ept, p, _ = _ept_and_p(func, inputs, kwargs)
        pp = np.power(p / orc.p0, orc.kappa)
        c_te = np.power(273.16 / (ept * pp), orc.LAMBDA)
        d = 1.0 / (0.1859e-5 * p + 0.6512)
        near = np.abs(c_te - d) <= thresh * d
        near |= np.abs(c_te - 1.0) <= thresh
        near |= np.abs(c_te - 0.4) <= thresh * 0.4
    return near

"""NumPy restatement of `earthkit.meteo.wind.array.w_from_omega` -- TEST INFRASTRUCTURE (the oracle).

Only the function SURVEY.md section 8f names.  Cites /root/reference/src/earthkit/meteo/wind/array/wind.py.
Pinned by tests/golden/wind_golden.npz (recorded from the reference by tests/golden/gen_golden_wind.py) and by
the reference's own known-answer vector (tests/wind/test_wind.py:183-194 there)."""
import numpy as np

Rd = 287.0597  # constants/constants.py:24
g = 9.80665    # constants/constants.py:53


def w_from_omega(omega, t, p):  # wind.py:192-222
    omega, t, p = (np.asarray(x) for x in (omega, t, p))
    return (-Rd / g) * (omega * t / p)

"""Seeded synthetic atmosphere generator -- TEST INFRASTRUCTURE.

Produces physically plausible (t, q, p) samples laid out ``[level, point]``
(level-major, C-contiguous).  Used by the tests, by ``gen_golden.py`` and by the
``cpu_baseline`` leg of ``bench.py``.  The timed GPU run generates its inputs
on the device with the same distribution (``ekm_synth_fill_*``); bit-equality
between the two generators is not needed because parity is always checked on
inputs copied from where they were generated.
"""

import numpy as np

from . import thermo_oracle as orc

SEED = 20260313
NLEV = 137
P_TOP = 1000.0
P_SFC = 101325.0


def level_pressures(nlev=NLEV):
    """137 levels linear in pressure, 10 hPa ... 1013.25 hPa."""
    k = np.arange(nlev, dtype=np.float64)
    return P_TOP + (P_SFC - P_TOP) * k / (nlev - 1)


def standard_temperature(p):
    return np.maximum(288.15 * np.power(p / 101325.0, 0.190263), 216.65)


def make_fields(nlev, npts, dtype=np.float32, seed=SEED, p_mode="field", levels=None):
    """Return (t, q, p, p_levels) with t, q of shape (nlev, npts).

    p_mode "field": p has shape (nlev, npts), p_k * (1 + 0.05 u), u ~ U(-1, 1).
    p_mode "level": p is the (nlev,) level vector itself.
    `levels` selects which of the 137 standard levels the slab holds
    (default: nlev levels spread evenly over the column).
    """
    rng = np.random.default_rng(seed)
    pl_all = level_pressures()
    if levels is None:
        levels = np.linspace(0, NLEV - 1, nlev).round().astype(int)
    pl = pl_all[np.asarray(levels)]
    if p_mode == "field":
        p = pl[:, None] * (1.0 + 0.05 * rng.uniform(-1.0, 1.0, size=(nlev, npts)))
    else:
        p = np.broadcast_to(pl[:, None], (nlev, npts))
    t = standard_temperature(p) + rng.normal(0.0, 8.0, size=(nlev, npts))
    t = np.clip(t, 180.0, 330.0)
    rh = rng.uniform(1.0, 100.0, size=(nlev, npts))
    with np.errstate(all="ignore"):
        q = orc.specific_humidity_from_relative_humidity(t, rh, np.array(p))
    q = np.minimum(q, 0.04)
    q = np.where(np.isnan(q), 3e-6, q)
    t = t.astype(dtype)
    q = q.astype(dtype)
    if p_mode == "field":
        p_out = np.ascontiguousarray(p.astype(dtype))
    else:
        p_out = pl.astype(dtype)
    return np.ascontiguousarray(t), np.ascontiguousarray(q), p_out, pl.astype(dtype)

#!/usr/bin/env python3
"""What missing values cost the fp64 kernels: the fast first pass marks a point with a non-finite output for the
plain-double pass, and a masked field is full of legitimately NaN results.  Since round 5 a non-finite output that a NaN
INPUT explains stays as it is (csrc/ops.hpp::OpDeps).  16 levels x 1800 x 3600 fp64, device-resident; kernel time of the
six-output pipeline and of the Newton wet-bulb on
  * the clean benchmark slab,
  * the slab with 30 % of its rows masked (t and q NaN: a land / sea mask),
  * the slab with 30 % of its points masked at random (every wave holds a masked lane),
  * hybrid levels 0-15 of the IFS L137 table (1-100 Pa: tw is NaN there by the reference's own p - es < eps rule, with
    finite inputs -- no input explains it, the plain pass runs),
and the same with every lane forced through the plain pass (tuning parameter f64_plain) for scale.

    python tools/f64_nan_rate.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]

import ekm_hip  # noqa: E402
from ekm_hip import _ffi, thermo  # noqa: E402
from ekm_hip.vertical import hybrid_level_parameters  # noqa: E402

NLEV, INNER = 16, 1800 * 3600


def kernel_ms(fn, reps=5):
    lib = _ffi.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    _ffi.check(lib.ekm_event_create(0, C.byref(e0)))
    _ffi.check(lib.ekm_event_create(0, C.byref(e1)))
    for _ in range(2):
        for o in fn():
            o.free()
    ekm_hip.synchronize()
    _ffi.check(lib.ekm_event_record(0, e0, None))
    for _ in range(reps):
        for o in fn():
            o.free()
    _ffi.check(lib.ekm_event_record(0, e1, None))
    ekm_hip.synchronize()
    ms = C.c_float()
    _ffi.check(lib.ekm_event_elapsed_ms(0, e0, e1, C.byref(ms)))
    return ms.value / reps


def main():
    lib = _ffi.lib()
    n = NLEV * INNER
    t, q, p = (ekm_hip.DeviceArray.empty((NLEV, INNER), np.float64) for _ in range(3))
    _ffi.check(lib.ekm_synth_fill_f64(0, None, t.ptr, q.ptr, p.ptr, 100 * INNER, n, INNER, 137, 20260313))  # levels 100-115
    ekm_hip.synchronize()
    rng = np.random.default_rng(3)
    th, qh = t.to_host(), q.to_host()
    rows = th.reshape(NLEV * 1800, 3600)
    mask_rows = np.zeros(NLEV * 1800, bool)
    start = rng.integers(0, NLEV * 1800 - 60, 160)
    for s in start:
        mask_rows[s:s + 54] = True  # contiguous bands: ~30 % of the rows
    tr, qr = rows.copy(), qh.reshape(rows.shape).copy()
    tr[mask_rows], qr[mask_rows] = np.nan, np.nan
    pts = rng.random(n) < 0.3
    tp, qp = th.ravel().copy(), qh.ravel().copy()
    tp[pts], qp[pts] = np.nan, np.nan
    A, B = hybrid_level_parameters(137)
    sp = ekm_hip.to_device(101325.0 * (1.0 - 0.35 * rng.random(INNER) ** 3))
    hyb = ekm_hip.HybridPressure(A[:NLEV + 1], B[:NLEV + 1], sp)
    cases = [("clean slab", t, q, p),
             (f"{mask_rows.mean():.0%} of the rows masked", ekm_hip.to_device(tr.reshape(NLEV, INNER)), ekm_hip.to_device(qr.reshape(NLEV, INNER)), p),
             (f"{pts.mean():.0%} of the points masked at random", ekm_hip.to_device(tp.reshape(NLEV, INNER)), ekm_hip.to_device(qp.reshape(NLEV, INNER)), p),
             ("hybrid levels 0-15 (1-100 Pa, finite inputs, tw NaN by the reference's rule)", t, q, hyb)]
    for name, ta, qa, pa in cases:
        line = f"{name:80s}"
        for wl, fn in (("P5", lambda: thermo.pipeline_full(ta, qa, pa)),
                       ("wet-bulb newton", lambda: (thermo.wet_bulb_temperature_from_specific_humidity(ta, qa, pa, t_method="newton"),))):
            _ffi.check(lib.ekm_set_tuning_param(b"f64_plain", 0))
            ms = kernel_ms(fn)
            _ffi.check(lib.ekm_set_tuning_param(b"f64_plain", 1))
            ms_plain = kernel_ms(fn, reps=2)
            _ffi.check(lib.ekm_set_tuning_param(b"f64_plain", 0))
            line += f"  {wl} {ms:7.3f} ms (every lane plain: {ms_plain:7.3f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()

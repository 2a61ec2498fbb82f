#!/usr/bin/env python3
"""Kernel time of potential_temperature for the operand patterns that do NOT take the aligned full-field kernels
(map_bcast / unaligned map_levels): how far each is from the aligned case.  64 levels x 1800 x 3600 fp32, device-resident.

    python tools/odd_shapes_rate.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]

import ekm_hip  # noqa: E402
from ekm_hip import _ffi, thermo  # noqa: E402


def kernel_ms(fn, reps=10):
    lib = _ffi.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    _ffi.check(lib.ekm_event_create(0, C.byref(e0)))
    _ffi.check(lib.ekm_event_create(0, C.byref(e1)))
    for _ in range(3):
        fn()
    ekm_hip.synchronize()
    _ffi.check(lib.ekm_event_record(0, e0, None))
    for _ in range(reps):
        fn()
    _ffi.check(lib.ekm_event_record(0, e1, None))
    ekm_hip.synchronize()
    ms = C.c_float()
    _ffi.check(lib.ekm_event_elapsed_ms(0, e0, e1, C.byref(ms)))
    return ms.value / reps


def main():
    nlev, nlat, nlon = 64, 1800, 3600
    n = nlev * nlat * nlon
    rng = np.random.default_rng(0)
    t = ekm_hip.to_device((250.0 + 50.0 * rng.random(n + 8, dtype=np.float32)))
    p = ekm_hip.to_device((3e4 + 7e4 * rng.random(n + 8, dtype=np.float32)))
    lev = ekm_hip.to_device(np.linspace(1e3, 1e5, nlev, dtype=np.float32))
    lon = ekm_hip.to_device(np.linspace(9e4, 1e5, nlon, dtype=np.float32))
    f3 = lambda a, off=0, m=n: a.flat_slice(off, off + m)  # noqa: E731
    cases = [
        ("aligned fields (map_fields)", lambda: thermo.potential_temperature(f3(t), f3(p)), n, 12),
        ("fields 4 B off 16-B alignment", lambda: thermo.potential_temperature(f3(t, 1), f3(p, 1)), n, 12),
        ("fields, n not a multiple of 4", lambda: thermo.potential_temperature(f3(t, 0, n - 3), f3(p, 0, n - 3)), n - 3, 12),
        ("p a level vector (137,1,1)-style", lambda: thermo.potential_temperature(f3(t).reshape(nlev, nlat, nlon), lev.reshape(nlev, 1, 1)), n, 8),
        ("p a level vector, field 4 B off alignment", lambda: thermo.potential_temperature(f3(t, 1).reshape(nlev, nlat, nlon), lev.reshape(nlev, 1, 1)), n, 8),
        ("p a level vector, rows of odd length (1799 x 3601)", lambda: thermo.potential_temperature(f3(t, 0, nlev * 1799 * 3601).reshape(nlev, 1799, 3601), lev.reshape(nlev, 1, 1)), nlev * 1799 * 3601, 8),
        ("p a scalar (0-d DeviceArray)", lambda: thermo.potential_temperature(f3(t), ekm_hip.to_device(np.float32(85000.0))), n, 8),
        ("p along the trailing axis (nlon,)", lambda: thermo.potential_temperature(f3(t).reshape(nlev, nlat, nlon), lon), n, 8),
        ("t a level vector, p a field", lambda: thermo.potential_temperature(lev.reshape(nlev, 1, 1), f3(p).reshape(nlev, nlat, nlon)), n, 8),
    ]
    base = None
    for name, fn, pts, bpp in cases:
        ms = kernel_ms(fn)
        gbs = bpp * pts / ms * 1e-6
        base = base or gbs
        print(f"{name:44s} {ms:7.3f} ms  {gbs:7.1f} GB/s  ({gbs / 8000:.3f} of 8 TB/s, {gbs / base:.2f} of the aligned case)", flush=True)


if __name__ == "__main__":
    main()

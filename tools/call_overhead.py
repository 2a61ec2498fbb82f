#!/usr/bin/env python3
"""Host-side cost of one thermo call on device-resident operands (BASELINE config 2's shape: 721x1440 fp64), where the
kernel is ~10 us and the Python layer decides the latency.

    python tools/call_overhead.py [--calls 4000] [--profile]

Prints the wall time per call, the kernel's own time (HIP events over the same loop), and with --profile the cProfile
table of the loop (top 25 by own time)."""
import argparse
import cProfile
import ctypes as C
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]

import ekm_hip  # noqa: E402
from ekm_hip import _ffi, thermo  # noqa: E402
from oracle import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=4000)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"])
    a = ap.parse_args()
    dt = np.float64 if a.dtype == "f64" else np.float32
    t, q, p, _ = synthetic.make_fields(1, 721 * 1440, dtype=dt, seed=2)
    d = [ekm_hip.to_device(x.reshape(721, 1440)) for x in (t, q, p)]
    cases = {
        "relative_humidity_from_specific_humidity(t, q, p)": lambda: thermo.relative_humidity_from_specific_humidity(*d),
        "potential_temperature(t, p)": lambda: thermo.potential_temperature(d[0], d[2]),
        "potential_temperature(t, 85000.0)": lambda: thermo.potential_temperature(d[0], 85000.0),
        "saturation_vapour_pressure(t)": lambda: thermo.saturation_vapour_pressure(d[0]),
    }
    lib = _ffi.lib()
    for name, fn in cases.items():
        for _ in range(50):
            fn()
        ekm_hip.synchronize()
        e0, e1 = C.c_void_p(), C.c_void_p()
        _ffi.check(lib.ekm_event_create(0, C.byref(e0)))
        _ffi.check(lib.ekm_event_create(0, C.byref(e1)))
        _ffi.check(lib.ekm_event_record(0, e0, None))
        t0 = time.perf_counter()
        for _ in range(a.calls):
            fn()
        t_issue = time.perf_counter() - t0
        _ffi.check(lib.ekm_event_record(0, e1, None))
        ekm_hip.synchronize()
        t_all = time.perf_counter() - t0
        ms = C.c_float()
        _ffi.check(lib.ekm_event_elapsed_ms(0, e0, e1, C.byref(ms)))
        print(f"{name:52s} host issue {1e6 * t_issue / a.calls:7.2f} us/call   wall {1e6 * t_all / a.calls:7.2f} us/call   "
              f"device span {1e3 * ms.value / a.calls:7.2f} us/call", flush=True)
    # the same four calls recorded once (ekm_hip.graph) and replayed
    with ekm_hip.graph() as g:
        keep = [fn() for name, fn in cases.items() if "85000" not in name]  # (a Python scalar operand is an upload)
    for _ in range(50):
        g.launch()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.calls):
        g.launch()
    t_issue = time.perf_counter() - t0
    g.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{'graph replay of the 3 DeviceArray-only calls above':52s} host issue {1e6 * t_issue / a.calls:7.2f} us/replay wall {1e6 * t_all / a.calls:7.2f} us/replay",
          flush=True)
    g.close()
    del keep
    if a.profile:
        fn = cases["relative_humidity_from_specific_humidity(t, q, p)"]
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(a.calls):
            fn()
        pr.disable()
        ekm_hip.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()

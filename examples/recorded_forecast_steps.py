#!/usr/bin/env python3
"""The same diagnostics for every step of a forecast, on a single-level field: record the calls once, replay them per step.

On a 721 x 1440 field a thermo kernel runs for a few microseconds -- less than the Python call that launches it -- so a
loop of such calls is bound by the host.  `ekm_hip.graph()` records the calls of one step into a HIP graph; each further
step uploads the new fields into the SAME device arrays and replays the graph with one launch.  Run from the repository root:

    python examples/recorded_forecast_steps.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "earthkit-meteo_amd"))

import ekm_hip  # noqa: E402
from ekm_hip import thermo  # noqa: E402


def diagnostics(t, q, p):
    """2 m-style diagnostics with the earthkit-meteo signatures; works on NumPy arrays and on DeviceArrays."""
    return (thermo.relative_humidity_from_specific_humidity(t, q, p), thermo.dewpoint_from_specific_humidity(q, p),
            thermo.ept_from_specific_humidity(t, q, p), thermo.wet_bulb_temperature_from_specific_humidity(t, q, p))


def main(nlat=721, nlon=1440, steps=24):
    rng = np.random.default_rng(1)
    p = (101325.0 * (1.0 - 0.25 * rng.random((nlat, nlon)) ** 3)).astype(np.float32)
    t0 = (288.0 - 40.0 * np.abs(np.linspace(-1, 1, nlat))[:, None] + rng.normal(0, 3, (nlat, nlon))).astype(np.float32)
    q0 = np.clip(0.012 * np.exp((t0 - 300.0) / 12.0), 1e-5, None).astype(np.float32)
    fields = [((t0 + 0.2 * k).astype(np.float32), (q0 * (1.0 + 0.01 * k)).astype(np.float32)) for k in range(steps)]

    d_t, d_q, d_p = ekm_hip.to_device(t0), ekm_hip.to_device(q0), ekm_hip.to_device(p)
    with ekm_hip.graph() as g:                      # nothing runs in here: the four launches are recorded
        outs = diagnostics(d_t, d_q, d_p)
    results = []
    for t, q in fields:
        d_t.copy_from_host(t)                       # same device arrays, new contents
        d_q.copy_from_host(q)
        g.launch()                                  # ordered after the uploads, asynchronous
        results.append([o.to_host() for o in outs])  # ordered after the launch

    # what the recording saves: the launches alone, eager against replayed
    ekm_hip.synchronize()
    tick = time.perf_counter()
    for _ in range(200):
        diagnostics(d_t, d_q, d_p)
    ekm_hip.synchronize()
    eager = (time.perf_counter() - tick) / 200
    tick = time.perf_counter()
    for _ in range(200):
        g.launch()
    g.synchronize()
    replay = (time.perf_counter() - tick) / 200
    print(f"{steps} steps of 4 diagnostics on {nlat} x {nlon}: per step eager {1e6 * eager:.0f} us, replayed {1e6 * replay:.0f} us")
    g.close()
    return fields, results


if __name__ == "__main__":
    main()

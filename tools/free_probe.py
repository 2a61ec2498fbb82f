#!/usr/bin/env python3
"""Round 6: what returning tens of GB of device memory to the driver does to host<->device copies afterwards (bench.py's
`end_to_end` ran at half rate whenever its buffer-set leg -- 64 GB allocated, written, freed -- came first, and so did the next
process on the device).  Copy rate before, then every half second after hipFree of N GB that kernels have written."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np

import ekm_hip
from ekm_hip import thermo

n = 8 * 1800 * 3600
h = np.random.default_rng(0).random(n).astype(np.float32)
out = np.empty(n, np.float32)
out.fill(0)
d = ekm_hip.to_device(h)


def rate():
    t0 = time.perf_counter()
    ekm_hip.DeviceArray.from_host(h).free() if False else None
    t0 = time.perf_counter()
    d2 = ekm_hip.to_device(h)
    ekm_hip.synchronize()
    up = h.nbytes / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    d2.to_host(out=out)
    dn = h.nbytes / (time.perf_counter() - t0) / 1e9
    d2.free()
    return up, dn


for gb, written in ((16, True), (64, True), (64, False)):
    print(f"--- {gb} GB allocated{', written by a kernel,' if written else ' (never touched)'} then returned to the driver", flush=True)
    print("before: h2d %.1f d2h %.1f GB/s" % rate(), flush=True)
    F = 1800 * 3600 * 137
    arrs = [ekm_hip.DeviceArray.empty((F,), np.float32) for _ in range(int(gb / 3.55))]
    if written:
        for i in range(0, len(arrs) - 1, 2):
            thermo.celsius_to_kelvin(arrs[i]).free()
        ekm_hip.synchronize()
    print("allocated: h2d %.1f d2h %.1f GB/s" % rate(), flush=True)
    for a in arrs:
        a.free()
    t0 = time.perf_counter()
    ekm_hip.empty_cache()
    d = ekm_hip.to_device(h)
    print(f"empty_cache() took {time.perf_counter() - t0:.2f} s", flush=True)
    t0 = time.perf_counter()
    last = None
    while time.perf_counter() - t0 < 20:
        r = rate()
        tag = "h2d %.1f d2h %.1f" % r
        if last is None or abs(r[0] - last) > 5:
            print(f"  +{time.perf_counter() - t0:5.1f} s: {tag} GB/s", flush=True)
            last = r[0]
        time.sleep(0.25)
    print(f"  +{time.perf_counter() - t0:5.1f} s: h2d %.1f d2h %.1f GB/s (end)" % rate(), flush=True)

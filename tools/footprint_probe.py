#!/usr/bin/env python3
"""Round 6: does a plain host<->device copy slow down as the process's VRAM footprint grows?  (bench.py's `end_to_end` phases ran at
29.8 GB/s after its three buffer sets -- 96 GB -- had been allocated, 56 GB/s at process start.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np

import ekm_hip

n = 8 * 1800 * 3600
h = np.random.default_rng(0).random(n).astype(np.float32)
out = np.empty(n, np.float32)
out.fill(0)


def rates(tag):
    b_up = b_dn = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        d = ekm_hip.to_device(h)
        ekm_hip.synchronize()
        b_up = min(b_up, time.perf_counter() - t0)
        t0 = time.perf_counter()
        d.to_host(out=out)
        b_dn = min(b_dn, time.perf_counter() - t0)
        ptr = d.ptr
        d.free()
    st = ekm_hip.memory_stats()
    print(f"{tag:58s} h2d {h.nbytes / b_up / 1e9:5.1f} GB/s  d2h {h.nbytes / b_dn / 1e9:5.1f} GB/s   live {st['live_bytes'] / 1e9:6.1f} GB cached {st['cached_bytes'] / 1e9:6.1f} GB  block at {ptr:#x}",
          flush=True)


rates("process start")
F = 137 * 1800 * 3600
sets = []
for k in range(3):
    sets.append([ekm_hip.DeviceArray.empty((F,), np.float32) for _ in range(9)])
    rates(f"after allocating buffer set {k + 1} (9 x 3.55 GB each)")
from ekm_hip import thermo
o = thermo.pipeline_full(sets[0][0], sets[0][1], sets[0][2])
ekm_hip.synchronize()
rates("after a six-output launch on set 1")
for x in o:
    x.free()
for s in sets[1:]:
    for a in s:
        a.free()
rates("sets 2 and 3 freed to the block cache")
ekm_hip.empty_cache()
rates("after empty_cache()")
for a in sets[0]:
    a.free()
ekm_hip.empty_cache()
rates("everything freed")

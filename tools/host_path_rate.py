#!/usr/bin/env python3
"""PCIe-inclusive rate of the NumPy-in / NumPy-out path (reported beside, never as, the HBM-resident
`value`): P3, theta and P5 on an 8-level and a 32-level slab of the benchmark field (1800 x 3600 fp32 points per
level = 26 MB per level and array), best of 5 calls: results in ordinary pageable arrays (EKM_PINNED_RESULTS=0), results
in pooled pinned memory (the default), and the caller's inputs in pinned memory too (ekm_hip.pinned_empty).  (Round 3
also timed three routes that were removed in round 4: profiles/r03_host_path_rate.txt.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import ekm_hip  # noqa: E402
from ekm_hip import _engine, thermo  # noqa: E402
from oracle import synthetic  # noqa: E402

LEVELS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 32]   # e.g. "1,2,4" for mid-size calls
if len(sys.argv) > 2:  # input bytes from which a call is streamed in slices (MB; the library's default: _engine._STREAM_BYTES)
    _engine._STREAM_BYTES = int(sys.argv[2]) << 20
for nlev in LEVELS:
    t, q, p, _ = synthetic.make_fields(nlev, 1800 * 3600, dtype=np.float32, seed=3)
    n = t.size
    for mode in ("pageable", "pooled", "pooled+in"):
        _engine._PINNED_OUT = mode.startswith("pooled")
        args3 = (t, q, p)
        if mode == "pooled+in":
            args3 = [ekm_hip.pinned_empty(a.shape, a.dtype) for a in (t, q, p)]
            for dst, src in zip(args3, (t, q, p)):
                dst[...] = src
        T, Q, P = args3
        for name, fn, args, nio in (("pipeline_svp_td_rh", thermo.pipeline_svp_td_rh, (T, Q, P), 6),
                                    ("potential_temperature", thermo.potential_temperature, (T, P), 3),
                                    ("pipeline_full", thermo.pipeline_full, (T, Q, P), 9)):
            best, res = 1e9, None
            for _ in range(5):
                res = None  # the previous result is released OUTSIDE the timed region (munmap of 600 MB costs ~30 ms by itself)
                t0 = time.perf_counter()
                res = fn(*args)
                best = min(best, time.perf_counter() - t0)
            res = None
            print(f"{mode:8s} {name:24s} {nlev:3d} levels, {nio} arrays x {t.nbytes / 1e6:.0f} MB over PCIe: "
                  f"{best * 1e3:7.1f} ms per call = {n / best / 1e9:5.2f} G grid-points/s, {nio * t.nbytes / best / 1e9:5.1f} GB/s "
                  f"both directions together", flush=True)
print(ekm_hip.memory_stats())

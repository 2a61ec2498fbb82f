#!/usr/bin/env python3
"""Every C-ABI call of one streamed NumPy-in / NumPy-out call (P3 on 8 levels), with the thread it came from and when:
the library handle is wrapped for one call.  This is how round 5 saw that a slice's upload and the previous slice's
download each ran at half rate while they overlapped (0.88 ms per 26 MB instead of 0.46).

    python tools/trace_host_calls.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import thermo, _ffi
from oracle import synthetic
lib = _ffi.lib()
log = []
class Wrap:
    def __init__(self, real): self._real = real
    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in ("ekm_h2d", "ekm_d2h", "ekm_stream_sync", "ekm_malloc", "ekm_event_record", "ekm_stream_wait_event", "ekm_pipeline_svp_td_rh_f32", "ekm_host_alloc"):
            def w(*a):
                t0 = time.perf_counter(); r = fn(*a); t1 = time.perf_counter()
                log.append((threading.current_thread().name, name, a[3] if name in ("ekm_h2d", "ekm_d2h") else 0, t0, t1)); return r
            return w
        return fn
t, q, p, _ = synthetic.make_fields(8, 1800*3600, dtype=np.float32, seed=3)
for i in range(3):
    r = None; r = thermo.pipeline_svp_td_rh(t, q, p)
_ffi._lib = Wrap(lib)
r = None
t0 = time.perf_counter(); r = thermo.pipeline_svp_td_rh(t, q, p); tt = time.perf_counter() - t0
_ffi._lib = lib
print("call", tt * 1e3, "ms")
z = min(e[3] for e in log)
for th, name, nb, a, b in sorted(log, key=lambda e: e[3]):
    if (b - a) > 20e-6:
        print(f"{th[-12:]:12s} {name:28s} {nb >> 20:4d} MB  {1e3 * (a - z):7.2f} -> {1e3 * (b - z):7.2f}  ({1e3 * (b - a):.3f} ms{', %.1f GB/s' % (nb / (b - a) / 1e9) if nb else ''})")

import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import thermo, _ffi, _engine
nlev, inner = 8, 1800*3600
rng = np.random.default_rng(0)
t = (250 + 30*rng.random((nlev, inner))).astype(np.float32)
q = (0.001 + 0.01*rng.random((nlev, inner))).astype(np.float32)
p = (50000 + 50000*rng.random((nlev, inner))).astype(np.float32)
f = lambda: thermo.pipeline_svp_td_rh(t, q, p)
f()
for mode in ("discard", "hold"):
    ts = []
    r = None
    for _ in range(5):
        t0 = time.perf_counter()
        if mode == "discard":
            f()
        else:
            r = f()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, ["%.1f" % x for x in ts])
for pre in (0, 8 << 20):
    _engine._PRETOUCH_BYTES = pre if pre else 1 << 60
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    print("pretouch", bool(pre), ["%.1f" % x for x in ts])

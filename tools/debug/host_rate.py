import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import thermo
nlev, inner = 8, 1800*3600
rng = np.random.default_rng(0)
t = (250 + 30*rng.random((nlev, inner))).astype(np.float32)
q = (0.001 + 0.01*rng.random((nlev, inner))).astype(np.float32)
p = (50000 + 50000*rng.random((nlev, inner))).astype(np.float32)
for name, f in (("p3", lambda: thermo.pipeline_svp_td_rh(t, q, p)), ("theta", lambda: thermo.potential_temperature(t, p))):
    f()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    n = t.size
    print(name, "NumPy in/out: %.1f ms -> %.2f Gpts/s" % (min(ts)*1e3, n/min(ts)/1e9))

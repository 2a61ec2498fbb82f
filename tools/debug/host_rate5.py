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
def bench(label, f):
    f(); ts=[]
    for _ in range(5):
        t0=time.perf_counter(); r=f(); ts.append((time.perf_counter()-t0)*1e3); del r
    print(label, "min %.1f ms" % min(ts), ["%.0f"%x for x in ts])
bench("single", lambda: thermo.pipeline_svp_td_rh(t,q,p))
for k in (2,4,8):
    def g():
        with ekm_hip.multi_gpu([0]*k):
            return thermo.pipeline_svp_td_rh(t,q,p)
    bench(f"{k} shards on one GPU", g)
ref = thermo.pipeline_svp_td_rh(t,q,p)
with ekm_hip.multi_gpu([0]*4):
    got = thermo.pipeline_svp_td_rh(t,q,p)
print("equal:", all(np.array_equal(a,b) for a,b in zip(ref,got)))

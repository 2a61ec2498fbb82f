import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import thermo, _ffi
lib=_ffi.lib()
nlev, inner = 8, 1800*3600
rng = np.random.default_rng(0)
t = (250 + 30*rng.random((nlev, inner))).astype(np.float32)
q = (0.001 + 0.01*rng.random((nlev, inner))).astype(np.float32)
p = (50000 + 50000*rng.random((nlev, inner))).astype(np.float32)
def T(f, n=3):
    b=1e9
    for _ in range(n):
        t0=time.perf_counter(); r=f(); b=min(b,time.perf_counter()-t0)
    return b*1e3
thermo.pipeline_svp_td_rh(t,q,p)
print("p3 numpy in/out total %.1f ms" % T(lambda: thermo.pipeline_svp_td_rh(t,q,p)))
d=[ekm_hip.to_device(a) for a in (t,q,p)]
print("3x to_device %.1f ms" % T(lambda: [ekm_hip.to_device(a) for a in (t,q,p)]))
outs=thermo.pipeline_svp_td_rh(*d); ekm_hip.synchronize()
print("kernel (device) %.2f ms" % T(lambda: (thermo.pipeline_svp_td_rh(*d), ekm_hip.synchronize())))
print("3x to_host (fresh np.empty) %.1f ms" % T(lambda: [o.to_host() for o in outs]))
pre=[np.empty_like(t) for _ in range(3)]
for a in pre: a.fill(0)
print("3x to_host (pre-touched out=) %.1f ms" % T(lambda: [o.to_host(out=b) for o,b in zip(outs,pre)]))
print("np.empty+fill 3x %.1f ms" % T(lambda: [np.empty_like(t).fill(0) for _ in range(3)]))

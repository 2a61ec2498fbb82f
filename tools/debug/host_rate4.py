import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import thermo, _ffi
from ekm_hip.device import DeviceArray
lib=_ffi.lib()
nlev, inner = 8, 1800*3600
rng = np.random.default_rng(0)
t = (250 + 30*rng.random((nlev, inner))).astype(np.float32)
q = (0.001 + 0.01*rng.random((nlev, inner))).astype(np.float32)
p = (50000 + 50000*rng.random((nlev, inner))).astype(np.float32)
thermo.pipeline_svp_td_rh(t,q,p)
tick=time.perf_counter
for it in range(4):
    T=[tick()]
    d=[DeviceArray.from_host(a) for a in (t,q,p)]; T.append(tick())
    res=thermo.pipeline_svp_td_rh(*d); ekm_hip.synchronize(); T.append(tick())
    outs=[np.empty(t.shape,np.float32) for _ in range(3)]; T.append(tick())
    for o,r in zip(outs,res): r.to_host(out=o)
    T.append(tick())
    for x in d+list(res): x.free()
    T.append(tick())
    del outs; T.append(tick())
    names=["H2D x3","kernel","np.empty x3","D2H x3 (fresh)","device free (cache)","del host outs (munmap)"]
    print(it, "  ".join(f"{n} {1e3*(b-a):.1f}" for n,a,b in zip(names,T,T[1:])))

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np, ctypes as C
np.seterr(all="ignore")
import ekm_hip
from ekm_hip import thermo, _ffi
from oracle import thermo_oracle as orc
nlev, inner = 137, 1800*3600
lib=_ffi.lib()
n=4*1024*1024
t,q,p=(ekm_hip.DeviceArray.empty((n,),np.float32) for _ in range(3))
_ffi.check(lib.ekm_synth_fill_f32(0,None,t.ptr,q.ptr,p.ptr,0,n,inner,nlev,20260313))
full=thermo.pipeline_full(t,q,p)
sep=thermo.wet_bulb_temperature_from_specific_humidity(t,q,p,t_method="newton")
a=full[5].to_host(); b=sep.to_host()
ht,hq,hp=t.to_host(),q.to_host(),p.to_host()
r=np.abs(a.astype(np.float64)-b)/np.abs(b)
bad=np.flatnonzero(r>1e-5)
print("n bad",bad.size, bad[:10])
want=orc.wet_bulb_temperature_from_specific_humidity(ht,hq,hp,"ifs","newton")
w64=orc.wet_bulb_temperature_from_specific_humidity(ht.astype(np.float64),hq.astype(np.float64),hp.astype(np.float64),"ifs","newton")
for i in bad[:10]:
    print(i, ht[i],hq[i],hp[i],"fused",a[i],"sep",b[i],"oracle32",want[i],"oracle64",w64[i])
ra=np.abs(a-want)/want; rb=np.abs(b-want)/want
print("fused vs oracle max",ra.max(),np.argmax(ra),"sep vs oracle max",rb.max(),np.argmax(rb))
# regime quantities in fp64 for bad points
for i in bad[:5]:
    T,Q,P=float(ht[i]),float(hq[i]),float(hp[i])
    ept=orc.ept_from_specific_humidity(np.array([T]),np.array([Q]),np.array([P]))[0]
    pp=(P/1e5)**orc.kappa; te=ept*pp; c=(273.16/te)**orc.LAMBDA; D=1/(0.1859e-5*P+0.6512)
    print(i,"c_te",c,"D",D,"c_te-1",c-1)

#!/usr/bin/env python3
"""Every combination of special operands (tests/_fuzz.py::SPECIAL) through the GPU library against the oracle, for the named
functions (default: all 94 cases): prints each point whose result differs (the test of the same name only asserts).

    python tools/special_probe.py [function ...]"""
import numpy as np, sys, itertools, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"earthkit-meteo_amd"), os.path.join(ROOT,"tests")]
import ekm_hip as ek
import _fuzz
from oracle import thermo_oracle as orc
np.seterr(all='ignore')
names=set(sys.argv[1:])
for dtype in (np.float32,np.float64):
  for func,keys,kw in _fuzz._case_table():
    if names and func not in names: continue
    ins=_fuzz.special_operands(keys,dtype)
    want=getattr(orc,func)(*[a.copy() for a in ins],**kw)
    got=getattr(ek.thermo,func)(*ins,**kw)
    wl=want if isinstance(want,tuple) else (want,)
    gl=got if isinstance(got,tuple) else (got,)
    for k,(w_,g_) in enumerate(zip(wl,gl)):
        w_=np.asarray(w_,dtype=np.float64); g_=np.asarray(g_,dtype=np.float64)
        same=(w_==g_)|(np.isnan(w_)&np.isnan(g_))
        both=np.isfinite(w_)&np.isfinite(g_)
        rel=np.zeros_like(w_); rel[both]=np.abs(g_[both]-w_[both])/np.maximum(np.abs(w_[both]),1e-300)
        bad=~same&~(both&(rel<=(1e-4 if dtype==np.float32 else 1e-7)))
        if kw.get('t_method')=='bisect':
            for kk,a in zip(keys,ins):
                if kk in ('p','q','w'): bad&=np.abs(a)<1e20
        for j in np.flatnonzero(bad)[:12]:
            print(dtype.__name__,func,kw,k,tuple(float(a[j]) for a in ins),'want',w_[j],'got',g_[j])

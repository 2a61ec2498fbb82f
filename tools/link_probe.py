#!/usr/bin/env python3
"""Round 6: does the PCIe link change speed / width, or a plain upload its rate, after seconds of HBM-only kernels or of idling?
(No: profiles/r06_host_path_rate.txt.)"""
import glob, os, sys, time
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import thermo
def link():
    out=[]
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            s=open(d+"/current_link_speed").read().strip(); w=open(d+"/current_link_width").read().strip()
            dpm=open(d+"/pp_dpm_pcie").read().strip().replace("\n"," | ") if os.path.exists(d+"/pp_dpm_pcie") else ""
            out.append(f"{os.path.basename(os.path.dirname(d))}: {s} x{w} {dpm}")
        except OSError: pass
    return "; ".join(out)
print("start", link(), flush=True)
n=8*1800*3600
h=np.random.default_rng(0).random(n).astype(np.float32)
d=ekm_hip.to_device(h)
def rate(k=3):
    b=1e9
    for _ in range(k):
        t0=time.perf_counter(); d2=ekm_hip.to_device(h); ekm_hip.synchronize(); b=min(b,time.perf_counter()-t0); d2.free()
    return h.nbytes/b/1e9
print("h2d at start %.1f GB/s"%rate(), link(), flush=True)
# busy the GPU with HBM-resident kernels for ~6 s, no PCIe traffic
big=[ekm_hip.DeviceArray.empty((60*1800*3600,), np.float32) for _ in range(2)]
t0=time.time()
while time.time()-t0<6:
    o=thermo.potential_temperature(big[0], big[1]); ekm_hip.synchronize(); o.free()
print("after 6 s of kernels:", link(), flush=True)
t1=time.time()
for i in range(12):
    print("  h2d %.1f GB/s at +%.3f s"%(rate(1), time.time()-t1), link() if i%4==0 else "", flush=True)
time.sleep(3)
print("after 3 s idle: h2d %.1f GB/s"%rate(1), link(), flush=True)
for i in range(6):
    print("  h2d %.1f GB/s"%rate(1), flush=True)

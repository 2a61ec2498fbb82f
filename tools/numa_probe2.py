#!/usr/bin/env python3
"""One process, like bench.py's: where do the host arrays live (NUMA node) and what does a host<->device copy get?"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip

def cpulist(txt):
    out = set()
    for part in txt.strip().split(","):
        if part:
            a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
    return out
nodes = {int(d.rsplit("node", 1)[1]): cpulist(open(d + "/cpulist").read()) for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))}
def where():
    cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1
    try:
        cpu = int(open("/proc/self/stat").read().split()[38])
    except Exception:
        pass
    return cpu, [n for n, c in nodes.items() if cpu in c]
print(subprocess.run(["rocm-smi", "--showtoponuma"], capture_output=True, text=True).stdout[-600:])
print("process runs on cpu/node", where(), flush=True)
n = 8 * 1800 * 3600
def rate_h2d(h, k=3):
    b = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); d = ekm_hip.to_device(h); ekm_hip.synchronize(); b = min(b, time.perf_counter() - t0); d.free()
    return h.nbytes / b / 1e9
def rate_d2h(d, out, k=3):
    b = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); d.to_host(out=out); b = min(b, time.perf_counter() - t0)
    return out.nbytes / b / 1e9
h0 = np.random.default_rng(0).random(n).astype(np.float32)
print("array made before any pinning: h2d %.1f GB/s" % rate_h2d(h0), flush=True)
big = [ekm_hip.DeviceArray.empty((137 * 1800 * 3600,), np.float32) for _ in range(9)]  # the benchmark's footprint
dsrc = ekm_hip.to_device(h0)
for node, cpus in nodes.items():
    os.sched_setaffinity(0, cpus)
    time.sleep(0.05)
    h = np.random.default_rng(1).random(n).astype(np.float32)   # first touch under this affinity
    out = np.empty(n, np.float32); out.fill(0)
    print(f"thread on node {node} {where()}: new array first-touched here: h2d {rate_h2d(h):.1f} GB/s; the OLD array from here {rate_h2d(h0):.1f}; d2h into a new array {rate_d2h(dsrc, out):.1f}", flush=True)
    got = dsrc.to_host()   # what bench.py's slab is: written by whoever does the staging copy
    print(f"   array produced by to_host(): h2d {rate_h2d(got):.1f} GB/s", flush=True)
os.sched_setaffinity(0, set().union(*nodes.values()))

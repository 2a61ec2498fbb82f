import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import ekm_hip
from ekm_hip import _engine, thermo
from oracle import synthetic
import inspect
for nlev in (8, 32):
    t, q, p, _ = synthetic.make_fields(nlev, 1800 * 3600, dtype=np.float32, seed=3)
    for lanes, pref in ((8, 256), (4, 256), (2, 256), (8, 1024), (4, 1024), (8, 64)):
        real = _engine.plan_slices
        def patched(rows, row_bytes, budget, max_lanes=8, min_slice=16 << 20, overhead=0, pref_slice=256 << 20, _l=lanes, _p=pref):
            return real(rows, row_bytes, budget, _l, min_slice, overhead, _p << 20)
        _engine.plan_slices = patched
        best = 1e9
        for _ in range(5):
            r = None
            t0 = time.perf_counter(); r = thermo.pipeline_svp_td_rh(t, q, p); best = min(best, time.perf_counter() - t0)
        _engine.plan_slices = real
        r = None
        rows = nlev
        print(f"{nlev} levels, lanes<={lanes}, pref slice {pref} MiB -> plan {real(rows, 6*t[0].nbytes, 1<<40, lanes, 16<<20, 0, pref<<20)}: {best*1e3:7.1f} ms = {6*t.nbytes/best/1e9:5.1f} GB/s", flush=True)

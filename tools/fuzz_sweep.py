#!/usr/bin/env python3
"""The wide-domain fuzz of tests/_fuzz.py with other seeds and more points than the suites run (they take one seed of
2^20 points): every theta_e method x {bisect, newton} x {fp32, fp64} x {from (theta_e, p), from (t, q, p)} on the GPU
against the oracle, judged by the same rules (tests/_fuzz.py::judge raises on the first real miss), plus the default walk
against the exact walk bit for bit.

    python tools/fuzz_sweep.py [--seeds 3] [--n 4194304]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tests")]

import ekm_hip  # noqa: E402
from ekm_hip import _ffi  # noqa: E402

import _fuzz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--n", type=int, default=1 << 22)
    a = ap.parse_args()
    np.seterr(all="ignore")
    lib = _ffi.lib()
    t0 = time.time()
    for s in range(a.seeds):
        seed = 1000 + 17 * s
        for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
            d = _fuzz.make(a.n, seed, dtype)
            dd = {k: ekm_hip.to_device(v) for k, v in d.items()}
            for func, keys, method, tm in _fuzz.CASES:
                ins = [dd[k] for k in keys]
                out = getattr(ekm_hip.thermo, func)(*ins, ept_method=method, t_method=tm)
                got = out.to_host()
                out.free()
                line = _fuzz.judge(func, keys, method, tm, tag, d, got)
                if tm == "bisect":
                    _ffi.check(lib.ekm_set_tuning_param(b"bisect_exact", 1))
                    ex = getattr(ekm_hip.thermo, func)(*ins, ept_method=method, t_method=tm)
                    _ffi.check(lib.ekm_set_tuning_param(b"bisect_exact", 0))
                    e = ex.to_host()
                    ex.free()
                    diff = int((~((got == e) | (np.isnan(got) & np.isnan(e)))).sum())
                    assert diff == 0, (func, method, tag, seed, diff)
                    line += "; default walk == exact walk on every point"
                print(f"seed {seed} {line}", flush=True)
            for v in dd.values():
                v.free()
    print(f"fuzz sweep: {a.seeds} seeds x {a.n} points x 24 cases: no real miss, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

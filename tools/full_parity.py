#!/usr/bin/env python3
"""Parity of the fused full pipeline on EVERY point of the benchmark field (3600 x 1800 x 137 fp32 =
887,760,000 grid points) against the NumPy oracle, level by level, oracle on all host cores.

    python tools/full_parity.py [--levels 137] [--out gpurun_out/full_parity.json]

For each of the six outputs: max relative error, points beyond 1e-4, NaN-pattern mismatches -- NO point is
excluded.  For the wet-bulb output additionally (oracle/census.py): the points beyond 1e-4 of the fp64
reference on the same fp32 inputs, how often the reference's own fp32 path misses its fp64 path, the number of
points whose Davies-Jones regime is decided by rounding (c_te within 1e-5 / 1e-6 of a threshold), and whether
any miss lies outside those bands.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
NAMES = ("theta", "es", "rh", "td", "theta_e", "tw")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    pool = mp.get_context("fork").Pool(cores)  # before HIP is initialised in this process

    import ekm_hip
    from ekm_hip import _ffi, thermo

    nlev, inner = a.levels, 1800 * 3600
    n = nlev * inner
    lib = _ffi.lib()
    t, q, p = (ekm_hip.DeviceArray.empty((n,), np.float32) for _ in range(3))
    _ffi.check(lib.ekm_synth_fill_f32(0, None, t.ptr, q.ptr, p.ptr, 0, n, inner, nlev, 20260313))
    outs = thermo.pipeline_full(t, q, p)
    ekm_hip.synchronize()

    from oracle import census

    parts = []
    t0 = time.time()
    chunk = inner // cores // 4 * 4
    for lev in range(nlev):
        base = lev * inner
        host = [x.flat_slice(base, base + inner).to_host() for x in (t, q, p) + tuple(outs)]
        jobs = []
        for lo in range(0, inner, chunk):
            hi = min(lo + chunk, inner)
            jobs.append(dict(kind="full", t=host[0][lo:hi], q=host[1][lo:hi], p=host[2][lo:hi],
                             got=[h[lo:hi] for h in host[3:]], tw_index=5))
        parts.append(census.merge(pool.map(census.job, jobs)))
        if lev % 8 == 0:
            tw = census.merge(parts)[5]
            print(f"level {lev + 1}/{nlev}  {time.time() - t0:.0f} s  tw: beyond 1e-4 so far {tw['over']} "
                  f"(reference fp32 vs fp64: {tw['reference_fp32_vs_fp64_over']})", flush=True)
    total = dict(zip(NAMES, census.merge(parts)))
    npts = n
    pool.close()
    res = dict(points=npts, levels=nlev, tolerance=1e-4, outputs=total, excluded_points=0,
               seconds=round(time.time() - t0, 1))
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

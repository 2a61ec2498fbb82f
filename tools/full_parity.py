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
    ap.add_argument("--pmode", default="field", choices=["field", "level", "hybrid"],
                    help="pressure as a full field, as the 137-level vector, or formed in the kernel from sp + A/B tables")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--workload", default="full", choices=["full", "bisect"],
                    help="full: the six-output pipeline; bisect: wet-bulb by the reference's default bisection (census in quanta)")
    a = ap.parse_args()
    dt = np.float32 if a.dtype == "f32" else np.float64
    tol = 1e-4 if a.dtype == "f32" else 1e-6
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    pool = mp.get_context("fork").Pool(cores)  # before HIP is initialised in this process

    import ekm_hip
    from ekm_hip import _ffi, thermo

    nlev, inner = a.levels, 1800 * 3600
    n = nlev * inner
    lib = _ffi.lib()
    t, q, p = (ekm_hip.DeviceArray.empty((n,), dt) for _ in range(3))
    plev_host = sp_host = Ah = Bh = None
    if a.pmode == "hybrid":
        from oracle import vertical_oracle as vo

        Ah, Bh = (x[137 - nlev:].astype(dt) for x in ekm_hip.vertical.hybrid_level_parameters(137))
        sp_host = (101325.0 * (1.0 - 0.35 * np.random.default_rng(20260313).random(inner) ** 3)).astype(dt)
        hp = ekm_hip.HybridPressure(Ah, Bh, ekm_hip.to_device(sp_host))
        # t, q drawn around the hybrid-level pressure (materialised once for the generator, then dropped)
        _ffi.check(getattr(lib, f"ekm_pressure_on_hybrid_levels_{a.dtype}")(0, None, ekm_hip.to_device(Ah).ptr, ekm_hip.to_device(Bh).ptr,
                                                         hp.sp.ptr, inner, nlev, None, None, 1, float(np.log(2)), p.ptr,
                                                         None, None, None))
        _ffi.check(getattr(lib, f"ekm_synth_fill_given_p_{a.dtype}")(0, None, t.ptr, q.ptr, p.ptr, 0, n, 20260313))
        ekm_hip.synchronize()
        outs = thermo.pipeline_full(t.reshape(nlev, inner), q.reshape(nlev, inner), hp)
    elif a.pmode == "level":
        plev = ekm_hip.DeviceArray.empty((nlev,), dt)
        _ffi.check(getattr(lib, f"ekm_synth_levels_{a.dtype}")(0, None, plev.ptr, nlev))
        _ffi.check(getattr(lib, f"ekm_synth_fill_{a.dtype}")(0, None, t.ptr, q.ptr, None, 0, n, inner, nlev, 20260313))
        plev_host = plev.to_host()
        outs = thermo.pipeline_full(t.reshape(nlev, inner), q.reshape(nlev, inner), plev.reshape(nlev, 1))
    elif a.workload == "bisect":
        _ffi.check(getattr(lib, f"ekm_synth_fill_{a.dtype}")(0, None, t.ptr, q.ptr, p.ptr, 0, n, inner, nlev, 20260313))
        outs = (thermo.wet_bulb_temperature_from_specific_humidity(t, q, p, ept_method="ifs", t_method="bisect"),)
    else:
        _ffi.check(getattr(lib, f"ekm_synth_fill_{a.dtype}")(0, None, t.ptr, q.ptr, p.ptr, 0, n, inner, nlev, 20260313))
        outs = thermo.pipeline_full(t, q, p)
    outs = tuple(o.ravel() for o in outs)
    ekm_hip.synchronize()

    from oracle import census

    parts, per_level = [], []
    t0 = time.time()
    chunk = inner // cores // 4 * 4
    for lev in range(nlev):
        base = lev * inner
        host = [x.flat_slice(base, base + inner).to_host() for x in (t, q, p) + tuple(outs)]
        if a.pmode == "level":  # the oracle gets the level's pressure as the reference would: broadcast over the level
            host[2] = np.full(inner, plev_host[lev], dt)
        elif a.pmode == "hybrid":  # ... or the hybrid definition evaluated by the (pinned) vertical oracle
            host[2] = np.ascontiguousarray(vo.pressure_on_hybrid_levels(Ah[lev:lev + 2], Bh[lev:lev + 2], sp_host)[0]
                                           .astype(dt))
        jobs = []
        for lo in range(0, inner, chunk):
            hi = min(lo + chunk, inner)
            if a.workload == "bisect":
                jobs.append(dict(t=host[0][lo:hi], q=host[1][lo:hi], p=host[2][lo:hi], got=host[3][lo:hi]))
            else:
                jobs.append(dict(kind="full", t=host[0][lo:hi], q=host[1][lo:hi], p=host[2][lo:hi],
                                 got=[h[lo:hi] for h in host[3:]], tw_index=5, tol=tol))
        if a.workload == "bisect":
            parts.append(census.merge(pool.map(census.bisect_job, jobs)))
            if lev % 8 == 0:
                b = census.merge(parts)[0]
                print(f"level {lev + 1}/{nlev}  {time.time() - t0:.0f} s  identical {b['identical']} of {b['n']}, differing "
                      f"{b['n'] - b['identical']} (unexplained {b['differ_unexplained']})", flush=True)
            continue
        parts.append(census.merge(pool.map(census.job, jobs)))
        per_level.append(dict(level=lev, p_mean=float(np.mean(host[2], dtype=np.float64)), tw_over=parts[-1][5]["over"],
                              tw_reference_fp32_vs_fp64_over=parts[-1][5]["reference_fp32_vs_fp64_over"],
                              tw_max_rel=parts[-1][5]["max_rel"]))
        if lev % 8 == 0:
            tw = census.merge(parts)[5]
            print(f"level {lev + 1}/{nlev}  {time.time() - t0:.0f} s  tw: beyond the bar so far {tw['over']} "
                  f"(reference fp32 vs fp64: {tw['reference_fp32_vs_fp64_over']})", flush=True)
    total = dict(zip(NAMES if a.workload == "full" else ("tw_bisect",), census.merge(parts)))
    npts = n
    pool.close()
    res = dict(points=npts, levels=nlev, workload=a.workload, p_mode=a.pmode, dtype=a.dtype, tolerance=tol if a.workload == "full" else "2 quanta of 120/4096 K", outputs=total, excluded_points=0,
               tw_per_level=[x for x in per_level if x["tw_over"] or x["tw_reference_fp32_vs_fp64_over"]],
               seconds=round(time.time() - t0, 1))
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Parity of the fused full pipeline on EVERY point of the benchmark field (3600 x 1800 x 137 fp32 =
887,760,000 grid points) against the NumPy oracle, level by level, oracle on all host cores.

    python tools/full_parity.py [--levels 137] [--out gpurun_out/full_parity.json]

For each of the six outputs: max relative error, points beyond 1e-4, NaN-pattern mismatches; for the
wet-bulb output the points whose Davies-Jones regime is decided by rounding (oracle/conditioning.py) are
counted separately and excluded, as everywhere else.
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


def check(job):
    from oracle import conditioning
    from oracle import thermo_oracle as orc

    t, q, p, got = job
    with np.errstate(all="ignore"):
        want = orc.pipeline_full(t, q, p)
        edge = conditioning.newton_regime_boundary("pipeline_full", [t, q, p], {}, 1e-5)
    res = {}
    for k, (name, g, w) in enumerate(zip(NAMES, got, want)):
        g = g.astype(np.float64)
        w = np.asarray(w, np.float64)
        keep = ~edge if name == "tw" else np.ones(g.size, bool)
        with np.errstate(all="ignore"):
            r = np.abs(g - w) / np.abs(w)
        r = np.where(np.isfinite(r), r, 0.0)
        res[name] = dict(max_rel=float(r[keep].max()) if keep.any() else 0.0, over=int((r[keep] > 1e-4).sum()),
                         nan_mismatch=int((np.isnan(g) != np.isnan(w))[keep].sum()), nan=int(np.isnan(w).sum()),
                         worst_edge=float(r[~keep].max()) if (~keep).any() else 0.0)
    res["edge"] = int(edge.sum())
    res["n"] = int(t.size)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cores = max(1, min(os.cpu_count() or 1, 16))
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

    total = {k: dict(max_rel=0.0, over=0, nan_mismatch=0, nan=0, worst_edge=0.0) for k in NAMES}
    edge = npts = 0
    t0 = time.time()
    chunk = inner // cores // 4 * 4
    for lev in range(nlev):
        base = lev * inner
        host = [x.flat_slice(base, base + inner).to_host() for x in (t, q, p) + tuple(outs)]
        jobs = []
        for lo in range(0, inner, chunk):
            hi = min(lo + chunk, inner)
            jobs.append((host[0][lo:hi], host[1][lo:hi], host[2][lo:hi], [h[lo:hi] for h in host[3:]]))
        for r in pool.map(check, jobs):
            edge += r["edge"]
            npts += r["n"]
            for k in NAMES:
                total[k]["max_rel"] = max(total[k]["max_rel"], r[k]["max_rel"])
                total[k]["worst_edge"] = max(total[k]["worst_edge"], r[k]["worst_edge"])
                for f in ("over", "nan_mismatch", "nan"):
                    total[k][f] += r[k][f]
        if lev % 8 == 0:
            print(f"level {lev + 1}/{nlev}  {time.time() - t0:.0f} s  tw max_rel so far {total['tw']['max_rel']:.2e}",
                  flush=True)
    pool.close()
    res = dict(points=npts, levels=nlev, tolerance=1e-4, outputs=total, regime_boundary_points_excluded_from_tw=edge,
               seconds=round(time.time() - t0, 1))
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

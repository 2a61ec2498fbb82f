#!/usr/bin/env python3
"""Diagnostic: per-level table of the hybrid-level P5 census and the details of every tw miss that the reference's own
amplification does not explain (tests/test_gpu_census.py::test_census_p5_hybrid_levels_all_137_levels)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tests")]
np.seterr(all="ignore")


def main():
    import ekm_hip as ek
    from ekm_hip import _ffi
    from oracle import census, conditioning, thermo_oracle as orc, vertical_oracle as vo
    import test_gpu_census as T

    NLEV, INNER, N3, SEED = T.NLEV, T.INNER, T.N3, T.SEED
    lib = _ffi.lib()
    t, q = (ek.DeviceArray.empty((N3,), np.float32) for _ in range(2))
    A, B = (x.astype(np.float32) for x in ek.vertical.hybrid_level_parameters(137))
    sp_host = (101325.0 * (1.0 - 0.35 * np.random.default_rng(SEED).random(INNER) ** 3)).astype(np.float32)
    d_sp, d_a, d_b = ek.to_device(sp_host), ek.to_device(A), ek.to_device(B)
    ptmp = ek.DeviceArray.empty((N3,), np.float32)
    _ffi.check(lib.ekm_pressure_on_hybrid_levels_f32(0, None, d_a.ptr, d_b.ptr, d_sp.ptr, INNER, NLEV, None, None, 1,
                                                      float(np.log(2)), ptmp.ptr, None, None, None))
    _ffi.check(lib.ekm_synth_fill_given_p_f32(0, None, t.ptr, q.ptr, ptmp.ptr, 0, N3, SEED))
    ek.synchronize()
    ptmp.free()
    outs = [o.ravel() for o in ek.thermo.pipeline_full(t.reshape(NLEV, INNER), q.reshape(NLEV, INNER), ek.HybridPressure(A, B, d_sp))]
    ek.synchronize()
    p_of = lambda lev: vo.pressure_on_hybrid_levels(A[lev:lev + 2], B[lev:lev + 2], sp_host)[0]  # noqa: E731
    nl = int(os.environ.get("DEBUG_LEVELS", "40"))
    total, per = census.run_levels(T._fetcher((t, q, None), outs, p_of), range(nl), INNER, np.float32, "full", 6, tw_index=5)
    for lev, r in per:
        e = r[5]
        if e["over"] or e["nan_mismatch"]:
            print(f"lev {lev:3d} p~{0.5 * (A[lev] + A[lev + 1]):9.2f} over {e['over']:7d} nanmm {e['nan_mismatch']:3d} ref_self {e['reference_fp32_vs_fp64_over']:7d} "
                  f"ref_nan {e['reference_fp32_vs_fp64_nan_mismatch']:3d} explained {e['over_explained_by_amplification']:7d} unexplained {e['over_unexplained']} worst {e['worst_over']:.2e}", flush=True)
    for lev, r in per:
        if not r[5]["over_unexplained"]:
            continue
        lo, hi = lev * INNER, (lev + 1) * INNER
        ht, hq = t.flat_slice(lo, hi).to_host(), q.flat_slice(lo, hi).to_host()
        hp = np.ascontiguousarray(p_of(lev).astype(np.float32))
        g = outs[5].flat_slice(lo, hi).to_host().astype(np.float64)
        w = orc.wet_bulb_temperature_from_specific_humidity(ht, hq, hp, "ifs", "newton").astype(np.float64)
        w64 = orc.wet_bulb_temperature_from_specific_humidity(ht.astype(np.float64), hq.astype(np.float64), hp.astype(np.float64), "ifs", "newton")
        r_ = np.abs(g - w) / np.abs(w)
        r_ = np.where(np.isfinite(r_), r_, 0.0)
        nanmm = np.isnan(g) != np.isnan(w)
        miss = np.flatnonzero((r_ > 1e-4) | nanmm)
        fin, edge = conditioning.newton_misses_explained(ht[miss], hq[miss], hp[miss], g[miss], w[miss], 1e-4)
        kap, cands = conditioning.newton_amplification(ht[miss], hq[miss], hp[miss], return_candidates=True)
        for n_, i in enumerate(miss):
            if fin[n_] or edge[n_]:
                continue
            print(f"UNEXPLAINED lev {lev} idx {i}: t={ht[i]!r} q={hq[i]!r} p={hp[i]!r} ours={g[i]!r} ref32={w[i]!r} ref64={w64[i]!r} "
                  f"r={r_[i]:.3e} kappa={kap[n_]:.3e} candidates={[float(f'{c:.6g}') for c in cands[:, n_]]}")
            for h in (1e-7, 1e-5, 1e-4):
                k2, c2 = conditioning.newton_amplification(ht[i:i + 1], hq[i:i + 1], hp[i:i + 1], h, return_candidates=True)
                print(f"   h={h:g}: kappa {k2[0]:.3e} candidates {[float(f'{c:.6g}') for c in c2[:, 0]]}")


if __name__ == "__main__":
    main()

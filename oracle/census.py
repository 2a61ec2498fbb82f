"""Error census of GPU outputs against the oracle on whole slabs -- TEST INFRASTRUCTURE.

Shared by tests/test_gpu_configs.py and tools/full_parity.py: for each output the maximum relative error, the
points beyond 1e-4 and the NaN-pattern mismatches; for the wet-bulb output additionally the census of the
Davies-Jones regime-boundary points (oracle/conditioning.py): how many points miss 1e-4 against the fp32
reference, against the fp64 reference on the same fp32 inputs, how often the reference's own fp32 path misses
its fp64 path, and whether any miss lies outside the band where the regime is decided by rounding.
"""
import numpy as np

from . import conditioning
from . import thermo_oracle as orc


def job(job):
    """Oracle + per-output error census of one slab: job = dict(kind, t, q, p, got=[arrays], tw_index).
    Runs in a host worker process (never touches HIP)."""
    np.seterr(all="ignore")
    t, q, p = job["t"], job["q"], job["p"]
    tol = job.get("tol", 1e-4)  # 1e-4 for fp32 outputs, 1e-6 for fp64
    call = {"p3": lambda a, b, c: orc.pipeline_svp_td_rh(a, b, c),
            "wetbulb": lambda a, b, c: (orc.wet_bulb_temperature_from_specific_humidity(a, b, c, "ifs", "newton"),),
            "full": lambda a, b, c: orc.pipeline_full(a, b, c)}[job["kind"]]
    want = call(t, q, p)
    res = []
    for k, (g, w) in enumerate(zip(job["got"], want)):
        g64, w64 = g.astype(np.float64), np.asarray(w, np.float64)
        r = np.abs(g64 - w64) / np.abs(w64)
        r = np.where(np.isfinite(r), r, 0.0)
        entry = dict(max_rel=float(r.max()), over=int((r > tol).sum()), nan_mismatch=int((np.isnan(g64) != np.isnan(w64)).sum()))
        if job["tw_index"] == k:
            # the same fp32 inputs through the oracle in fp64: the "true" one-Newton-step answer, and the census of
            # points whose Davies-Jones regime the reference itself decides by rounding
            w_true = call(*(x.astype(np.float64) for x in (t, q, p)))[k]
            r64 = np.abs(g64 - w_true) / np.abs(w_true)
            r64 = np.where(np.isfinite(r64), r64, 0.0)
            ref_self = np.abs(w64 - w_true) / np.abs(w_true)  # the reference's own fp32 vs fp64
            ref_self = np.where(np.isfinite(ref_self), ref_self, 0.0)
            band5 = conditioning.newton_regime_boundary("pipeline_full", [t, q, p], {}, 1e-5)
            band6 = conditioning.newton_regime_boundary("pipeline_full", [t, q, p], {}, 1e-6)
            entry.update(over_vs_fp64_oracle=int((r64 > tol).sum()), max_rel_vs_fp64_oracle=float(r64.max()),
                         reference_fp32_vs_fp64_over=int((ref_self > tol).sum()),
                         band_1e5=int(band5.sum()), band_1e6=int(band6.sum()),
                         over_outside_band_1e5=int((r[~band5] > tol).sum()),
                         over_outside_band_1e6=int((r[~band6] > tol).sum()),
                         over_and_reference_agrees_with_itself=int(((r > tol) & (ref_self <= tol)).sum()),
                         worst_over=float(r[r > tol].max()) if (r > tol).any() else 0.0)
        res.append(entry)
    return res


def bisect_job(job):
    """Bisection wet-bulb (the reference's default t_method) on one slab: job = dict(t, q, p, got).  The result is
    quantised to 120/4096 K, so the census is in quanta: how many points equal the fp32 reference bit for bit, how many
    sit one / two / more quanta away, NaN-pattern differences, and how many of the non-identical points are ones
    where the reference's own residual is below fp32 noise at some step (oracle/conditioning.py, threshold 3e-6) or
    where its fp32 and fp64 runs disagree."""
    np.seterr(all="ignore")
    t, q, p, got = job["t"], job["q"], job["p"], job["got"]
    want = orc.wet_bulb_temperature_from_specific_humidity(t, q, p, "ifs", "bisect")
    g64, w64 = got.astype(np.float64), np.asarray(want, np.float64)
    quantum = 120.0 / 4096.0
    nanmm = np.isnan(g64) != np.isnan(w64)
    d = np.abs(g64 - w64) / quantum
    d = np.where(np.isfinite(d), d, 0.0)
    differ = (d > 0) | nanmm
    res = dict(n=int(t.size), identical=int((~differ).sum()), one_quantum=int(((d > 0) & (d <= 1.0001)).sum()),
               two_quanta=int(((d > 1.0001) & (d <= 2.0001)).sum()), more=int((d > 2.0001).sum()),
               nan_mismatch=int(nanmm.sum()), max_quanta=float(d.max()))
    if differ.any():  # classify the differing points only (the fp64 reference run is the expensive part)
        idx = np.flatnonzero(differ)
        noisy = conditioning.bisect_sign_noise("wet_bulb_temperature_from_specific_humidity", [t[idx], q[idx], p[idx]], {}, 3e-6)
        w_true = orc.wet_bulb_temperature_from_specific_humidity(*(x[idx].astype(np.float64) for x in (t, q, p)), "ifs", "bisect")
        w32 = w64[idx]
        unstable = (np.isnan(w32) != np.isnan(w_true)) | (np.abs(w32 - w_true) > 0.5 * quantum)
        res.update(differ_sign_noise=int(noisy.sum()), differ_reference_unstable=int(unstable.sum()),
                   differ_unexplained=int((~(noisy | unstable)).sum()))
    else:
        res.update(differ_sign_noise=0, differ_reference_unstable=0, differ_unexplained=0)
    return [res]


def merge(parts):
    """Sum the counts and take the maxima of the per-slab results of `job`."""
    total = [dict() for _ in parts[0]]
    for part in parts:
        for k, e in enumerate(part):
            for key, v in e.items():
                total[k][key] = max(total[k].get(key, 0.0), v) if key.startswith(("max_", "worst")) else total[k].get(key, 0) + v
    return total

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


def merge(parts):
    """Sum the counts and take the maxima of the per-slab results of `job`."""
    total = [dict() for _ in parts[0]]
    for part in parts:
        for k, e in enumerate(part):
            for key, v in e.items():
                total[k][key] = max(total[k].get(key, 0.0), v) if key.startswith(("max_", "worst")) else total[k].get(key, 0) + v
    return total

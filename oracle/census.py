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
    # the "true" one-Newton-step wet-bulb: the same fp32 inputs through the oracle in fp64 (the wet-bulb call alone --
    # by definition the tw member of the fused compositions)
    tw64 = lambda a, b, c: orc.wet_bulb_temperature_from_specific_humidity(a, b, c, "ifs", "newton")  # noqa: E731
    res = []
    for k, (g, w) in enumerate(zip(job["got"], want)):
        g64, w64 = g.astype(np.float64), np.asarray(w, np.float64)
        r = np.abs(g64 - w64) / np.abs(w64)
        r = np.where(np.isfinite(r), r, 0.0)
        entry = dict(max_rel=float(r.max()), over=int((r > tol).sum()), nan_mismatch=int((np.isnan(g64) != np.isnan(w64)).sum()))
        if job["tw_index"] == k:
            # the same fp32 inputs through the oracle in fp64: the "true" one-Newton-step answer, and the census of
            # points whose Davies-Jones regime the reference itself decides by rounding
            w_true = tw64(*(x.astype(np.float64) for x in (t, q, p)))
            r64 = np.abs(g64 - w_true) / np.abs(w_true)
            r64 = np.where(np.isfinite(r64), r64, 0.0)
            ref_self = np.abs(w64 - w_true) / np.abs(w_true)  # the reference's own fp32 vs fp64
            ref_self = np.where(np.isfinite(ref_self), ref_self, 0.0)
            band5, band6 = conditioning.newton_regime_boundary("pipeline_full", [t, q, p], {}, (1e-5, 1e-6))
            entry.update(over_vs_fp64_oracle=int((r64 > tol).sum()), max_rel_vs_fp64_oracle=float(r64.max()),
                         reference_fp32_vs_fp64_over=int((ref_self > tol).sum()),
                         band_1e5=int(band5.sum()), band_1e6=int(band6.sum()),
                         over_outside_band_1e5=int((r[~band5] > tol).sum()),
                         over_outside_band_1e6=int((r[~band6] > tol).sum()),
                         over_and_reference_agrees_with_itself=int(((r > tol) & (ref_self <= tol)).sum()),
                         worst_over=float(r[r > tol].max()) if (r > tol).any() else 0.0,
                         reference_fp32_vs_fp64_nan_mismatch=int((np.isnan(w64) != np.isnan(w_true)).sum()))
            # every miss is classified by how strongly the reference's OWN algorithm amplifies input perturbations
            # there (oracle/conditioning.py::newton_misses_explained, fp64 oracle only): explained iff the deviation is
            # within 8 x kappa x 2^-24, or the point sits on a NaN / regime edge (kappa = inf) AND the value under test is
            # one of the outcomes that edge offers (the fp64 oracle at the point or at a 1e-6 perturbation of an input)
            nanmm = np.isnan(g64) != np.isnan(w64)
            miss = np.flatnonzero((r > tol) | nanmm)
            expl = np.zeros(miss.size, bool)
            expl_finite = expl_edge = expl
            if miss.size:
                expl_finite, expl_edge = conditioning.newton_misses_explained(t[miss], q[miss], p[miss], g64[miss], w64[miss], tol)
            # ... or it is a Davies-Jones regime tie that the fp32 reference's own rounding flipped: the point lies in
            # the 1e-5 regime band and the output under test sides with the fp64 reference.  The narrower class is
            # named first (a tie also lies "inside the span of the outcomes the edge offers" since round 5), except
            # where the finite-kappa bound already explains the point.
            flip = ~expl_finite & band5[miss] & ~nanmm[miss] & (r64[miss] <= tol)
            expl_edge = expl_edge & ~flip
            expl = expl_finite | expl_edge
            entry.update(over_explained_by_amplification=int(expl.sum()), over_explained_finite_kappa=int(expl_finite.sum()),
                         over_explained_on_a_nan_edge=int((expl_edge & ~expl_finite).sum()),
                         over_regime_flip_of_the_fp32_reference=int(flip.sum()),
                         over_unexplained=int((~(expl | flip)).sum()))
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
    method = job.get("method", "ifs")
    want = orc.wet_bulb_temperature_from_specific_humidity(t, q, p, method, "bisect")
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
        noisy, noise_t = conditioning.bisect_sign_noise("wet_bulb_temperature_from_specific_humidity",
                                                        [t[idx], q[idx], p[idx]], {"ept_method": method}, 3e-6, return_points=True)
        w_true = orc.wet_bulb_temperature_from_specific_humidity(*(x[idx].astype(np.float64) for x in (t, q, p)), method, "bisect")
        w32, g = w64[idx], g64[idx]
        unstable = (np.isnan(w32) != np.isnan(w_true)) | (np.abs(w32 - w_true) > 0.5 * quantum)
        # NO point is exempt from a check (tests/_compare.py::_assert_bisect): a differing value must be a NaN that the
        # fp32 or the fp64 reference also has, or a finite value within 2 quanta of the fp32 reference, of the fp64
        # reference, or of a lattice temperature at which the reference's own residual is rounding noise
        two = 2.0001 * quantum
        near = np.isfinite(g) & ((np.abs(g - w32) <= two) | (np.abs(g - w_true) <= two) |
                                 (np.nanmin(np.where(np.isnan(noise_t), np.inf, np.abs(noise_t - g[:, None])), axis=1) <= two))
        anchored = np.where(np.isnan(g), np.isnan(w32) | np.isnan(w_true), near)
        res.update(differ_sign_noise=int(noisy.sum()), differ_reference_unstable=int(unstable.sum()),
                   differ_unexplained=int((~(noisy | unstable)).sum()), differ_unanchored=int((~anchored).sum()))
    else:
        res.update(differ_sign_noise=0, differ_reference_unstable=0, differ_unexplained=0, differ_unanchored=0)
    return [res]


def merge(parts):
    """Sum the counts and take the maxima of the per-slab results of `job`."""
    total = [dict() for _ in parts[0]]
    for part in parts:
        for k, e in enumerate(part):
            for key, v in e.items():
                total[k][key] = max(total[k].get(key, 0.0), v) if key.startswith(("max_", "worst")) else total[k].get(key, 0) + v
    return total


# ---- whole-field driver: level by level through a pool of host processes, no pickling of the data -------------
_MAPS = {}


def _slot_view(path, nrows, n, dtype, mode="r"):
    key = (path, nrows, n, str(dtype), mode)
    if key not in _MAPS:
        _MAPS[key] = np.memmap(path, dtype=dtype, mode=mode, shape=(nrows, n))
    return _MAPS[key]


def slot_job(spec):
    """One piece [lo, hi) of one level held in a memory-mapped slot file (rows: t, q, p, outputs...): runs `job` /
    `bisect_job` on views of it.  Executed in the pool's worker processes (spawned: they never touch HIP)."""
    a = _slot_view(spec["path"], spec["nrows"], spec["n"], spec["dtype"])
    lo, hi = spec["lo"], spec["hi"]
    t, q, p = a[0, lo:hi], a[1, lo:hi], a[2, lo:hi]
    got = [a[3 + k, lo:hi] for k in range(spec["nrows"] - 3)]
    if spec["kind"].startswith("bisect"):  # "bisect" (ifs), "bisect:bolton35", "bisect:bolton39"
        return bisect_job(dict(t=t, q=q, p=p, got=got[0], method=(spec["kind"].split(":") + ["ifs"])[1]))
    return job(dict(kind=spec["kind"], t=t, q=q, p=p, got=got, tw_index=spec["tw_index"], tol=spec["tol"]))


def run_levels(fetch_level, levels, n, dtype, kind, nout, tw_index=None, tol=1e-4, workers=None, inflight=3,
               tmpdir="/dev/shm", progress=None):
    """Census of `levels` (an iterable of level indices) of a field whose levels hold `n` points each.

    `fetch_level(lev, rows)` fills rows[0..2] = t, q, p and rows[3..] = the outputs under test of level `lev`
    (rows: a (3 + nout, n) array in shared memory; the caller downloads straight into it).  Each level is cut into
    pieces for a pool of spawned host processes that map the same file -- nothing is pickled but the small specs
    and results -- and up to `inflight` levels are in the pool while the next one is being fetched.  Returns
    (merged totals as from `merge`, per-level list of merged results)."""
    import multiprocessing as mp
    import os

    dtype = np.dtype(dtype)
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0))))
    nrows = 3 + nout
    paths = [os.path.join(tmpdir, f"ekm_census_{os.getpid()}_{k}.bin") for k in range(inflight)]
    piece = -(-n // (2 * workers))
    piece = -(-piece // 64) * 64
    per_level, pending = [], []
    try:
        slots = [_slot_view(pth, nrows, n, dtype, "w+") for pth in paths]
        with mp.get_context("spawn").Pool(workers) as pool:
            for i, lev in enumerate(levels):
                k = i % inflight
                if len(pending) >= inflight:  # the slot about to be overwritten must have been consumed
                    lv, res = pending.pop(0)
                    per_level.append((lv, merge(res.get())))
                    if progress:
                        progress(lv, per_level)
                fetch_level(lev, slots[k])
                specs = [dict(path=paths[k], nrows=nrows, n=n, dtype=str(dtype), lo=lo, hi=min(lo + piece, n), kind=kind,
                              tw_index=tw_index, tol=tol) for lo in range(0, n, piece)]
                pending.append((lev, pool.map_async(slot_job, specs, chunksize=1)))
            for lv, res in pending:
                per_level.append((lv, merge(res.get())))
                if progress:
                    progress(lv, per_level)
    finally:
        for key in [k for k in _MAPS if k[0] in paths]:
            del _MAPS[key]
        for pth in paths:
            try:
                os.unlink(pth)
            except OSError:
                pass
    return merge([r for _, r in per_level]), per_level

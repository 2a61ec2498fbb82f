"""The whole-field census driver (oracle/census.py::run_levels) on the CPU: the level-by-level pool with its
memory-mapped slots must find an injected error, classify it, and leave nothing behind in /dev/shm.  (On the GPU box
the same driver checks all 887.76 M points of the benchmark field: tests/test_gpu_census.py.)"""
import glob
import os

import numpy as np

from oracle import census, synthetic
from oracle import thermo_oracle as orc

np.seterr(all="ignore")
N = 20000


def _fetch_full(lev, rows):
    t, q, p, _ = synthetic.make_fields(1, N, dtype=np.float32, seed=lev, levels=[lev])
    rows[0], rows[1], rows[2] = t[0], q[0], p[0]
    for k, o in enumerate(orc.pipeline_full(t[0], q[0], p[0])):
        rows[3 + k] = o
    if lev == 100:
        rows[8, 5] *= 1.001   # one wet-bulb value 1e-3 off on a well-conditioned point
        rows[4, 7] = np.nan   # one NaN that the reference does not have


def _fetch_bisect(lev, rows):
    t, q, p, _ = synthetic.make_fields(1, N, dtype=np.float32, seed=lev, levels=[lev])
    rows[0], rows[1], rows[2] = t[0], q[0], p[0]
    rows[3] = orc.wet_bulb_temperature_from_specific_humidity(t[0], q[0], p[0], "ifs", "bisect")
    if lev == 100:
        rows[3, 11] += 5 * 120.0 / 4096.0   # five quanta off: anchored to nothing


def test_run_levels_finds_and_classifies_injected_errors():
    total, per = census.run_levels(_fetch_full, [3, 100], N, np.float32, "full", 6, tw_index=5, workers=2)
    per = dict(per)
    assert [e["over"] for e in total] == [0, 0, 0, 0, 0, 1]
    assert total[1]["nan_mismatch"] == 1 and total[5]["nan_mismatch"] == 0
    tw = total[5]
    assert tw["over_unexplained"] == 1 and tw["over_explained_by_amplification"] == 0 and abs(tw["worst_over"] - 1e-3) < 1e-4
    assert per[3][5]["over"] == 0 and per[100][5]["over"] == 1
    total, _ = census.run_levels(_fetch_bisect, [3, 100], N, np.float32, "bisect", 1, workers=2)
    b = total[0]
    assert b["n"] == 2 * N and b["more"] == 1 and b["differ_unanchored"] == 1 and b["identical"] == 2 * N - 1
    assert not glob.glob(f"/dev/shm/ekm_census_{os.getpid()}_*")


def test_newton_amplification_separates_hpa_level_points_from_tropospheric_ones():
    """oracle/conditioning.py::newton_amplification: O(1) in the troposphere, 1e2-1e6 where es(tw) ~ p."""
    from oracle import conditioning

    t, q, p, _ = synthetic.make_fields(2, 2000, dtype=np.float32, seed=5, levels=[60, 130])
    k = conditioning.newton_amplification(t.ravel(), q.ravel(), p.ravel())
    assert np.isfinite(k).all() and np.median(k) < 10 and k.max() < 100
    tt = np.full(2000, 232.0) + np.linspace(-6, 6, 2000)
    k2 = conditioning.newton_amplification(tt, np.full(2000, 3e-6), np.full(2000, 16.0))
    assert np.median(k2[np.isfinite(k2)]) > 50


def test_a_miss_on_a_nan_edge_must_be_one_of_the_outcomes_the_edge_offers():
    """kappa = inf (a 1e-6 perturbation of an input changes the NaN-ness of the fp64 oracle) used to
    accept ANY value.  Now the value under test must be NaN where the oracle or one of its perturbed evaluations is, or
    a finite value near one of the finite ones -- garbage on such a point is a real miss."""
    from oracle import conditioning

    # 16 Pa, dry: find where the one-step Newton wet-bulb of the fp64 oracle turns NaN along t (the p - es < eps edge),
    # then sample within +-0.5 mK of it -- closer than the 1e-6 relative perturbation (0.23 mK) the rule probes with
    coarse = np.full(4000, 232.0) + np.linspace(-8, 8, 4000)
    base = orc.wet_bulb_temperature_from_specific_humidity(coarse, np.full(4000, 3e-6), np.full(4000, 16.0), "ifs", "newton")
    flips = np.flatnonzero(np.isnan(base[1:]) != np.isnan(base[:-1]))
    assert flips.size, "the sweep crosses no NaN edge: pick another column"
    lo, hi = coarse[flips[0]], coarse[flips[0] + 1]
    for _ in range(40):  # bisect the edge down to rounding
        mid = 0.5 * (lo + hi)
        v = orc.wet_bulb_temperature_from_specific_humidity(np.array([lo, mid]), np.full(2, 3e-6), np.full(2, 16.0), "ifs", "newton")
        lo, hi = (mid, hi) if np.isnan(v[0]) == np.isnan(v[1]) else (lo, mid)
    n = 2000
    t = lo + np.linspace(-5e-4, 5e-4, n)
    q, p = np.full(n, 3e-6), np.full(n, 16.0)
    kap, cands = conditioning.newton_amplification(t, q, p, return_candidates=True)
    edge = np.flatnonzero(np.isinf(kap) & np.isfinite(cands).any(axis=0) & np.isnan(cands).any(axis=0))
    assert edge.size > 0, "the sweep crosses no NaN edge: pick another column"
    te, qe, pe = t[edge], q[edge], p[edge]
    want = orc.wet_bulb_temperature_from_specific_humidity(te, qe, pe, "ifs", "newton")
    first_finite = np.array([c[np.isfinite(c)][0] for c in cands[:, edge].T])
    for got, ok in ((np.full(edge.size, np.nan), True),             # NaN: some outcome of the edge is NaN
                    (first_finite * (1 + 2e-5), True),              # a finite outcome of the edge (within the bar of the span)
                    (np.full(edge.size, 123.456), False),           # garbage
                    (first_finite + 50.0, False)):                  # a finite value far from every outcome
        fin, on_edge = conditioning.newton_misses_explained(te, qe, pe, got, want, 1e-4)
        assert not fin.any()                                        # kappa = inf: never through the finite-kappa rule
        assert on_edge.all() == ok and on_edge.any() == ok, (got[:3], on_edge[:8])

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

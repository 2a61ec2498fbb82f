#!/usr/bin/env python3
"""Golden vectors for `wind.w_from_omega`, recorded from the REFERENCE (build container only; stand-in for
the un-vendored earthkit-utils in tests/golden/_standin).  Writes tests/golden/wind_golden.npz: inputs and the
reference's outputs in fp64 and fp32, a level-vector broadcast case, and the reference's own known-answer
vector (tests/wind/test_wind.py:183-194 there).  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("EKM_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "_standin"), os.path.join(REF, "src"), ROOT]

from earthkit.meteo.wind import array as ref  # noqa: E402

from oracle import synthetic  # noqa: E402

np.seterr(all="ignore")


def main():
    store = {}
    rng = np.random.default_rng(20260313)
    t, q, p, pl = synthetic.make_fields(16, 257, dtype=np.float64, seed=7)
    omega = rng.normal(0.0, 2.0, size=t.shape)
    omega.ravel()[:6] = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-30]
    p_edge = p.copy()
    p_edge.ravel()[6:10] = [0.0, -1.0, np.inf, np.nan]
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        o, tt, pp = omega.astype(dt), t.astype(dt), p_edge.astype(dt)
        store[f"{tag}.in.omega"], store[f"{tag}.in.t"], store[f"{tag}.in.p"] = o, tt, pp
        store[f"{tag}.out.field"] = np.asarray(ref.w_from_omega(o, tt, pp))
        store[f"{tag}.in.plev"] = pl.astype(dt)
        store[f"{tag}.out.levmajor"] = np.asarray(ref.w_from_omega(o, tt, pl.astype(dt)[:, None]))
        store[f"{tag}.out.scalar_p"] = np.asarray(ref.w_from_omega(o, tt, dt(85000.0)))
    ko, kt, kp = np.array([1.2, 21.3]), np.array([285.6, 261.1]), np.array([1000.0, 850.0]) * 100.0
    store["kat.omega"], store["kat.t"], store["kat.p"] = ko, kt, kp
    store["kat.expected"] = np.array([-0.1003208031, -1.9152219066])  # the values in the reference's test
    store["kat.out"] = np.asarray(ref.w_from_omega(ko, kt, kp))
    store["scalar.out"] = np.asarray(ref.w_from_omega(1.2, 285.6, 100000.0))
    np.savez_compressed(os.path.join(HERE, "wind_golden.npz"), **store)
    print("wrote", len(store), "arrays")


if __name__ == "__main__":
    main()

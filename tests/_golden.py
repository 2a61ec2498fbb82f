"""Loader for the committed golden vectors (tests/golden/*.npz)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_G = None
_C = None


def golden():
    global _G
    if _G is None:
        _G = np.load(os.path.join(HERE, "golden", "thermo_golden.npz"))
    return _G


def ref_csv():
    global _C
    if _C is None:
        _C = np.load(os.path.join(HERE, "golden", "ref_csv.npz"))
    return _C


def manifest():
    return json.loads(bytes(golden()["manifest"]).decode())


def case_inputs(case):
    g = golden()
    return [g[f"{case['dataset']}.{case['dtype']}.in.{a}"] for a in case["args"]]


def case_outputs(case):
    g = golden()
    return [g[f"{case['id']}.out{i}"] for i in range(case["nout"])]


def max_rel(a, b):
    """max |a-b|/|b| over finite b (0 when both are 0); NaN/inf patterns must be checked separately."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    fin = np.isfinite(b) & np.isfinite(a)
    if not fin.any():
        return 0.0
    d = np.abs(a[fin] - b[fin])
    den = np.abs(b[fin])
    with np.errstate(all="ignore"):
        r = np.where(d == 0, 0.0, d / den)
    return float(r.max())


def same_nonfinite(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    inf = np.isinf(b)
    return np.array_equal(np.isinf(a), inf) and np.array_equal(a[inf], b[inf])

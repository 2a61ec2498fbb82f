#!/usr/bin/env python3
"""Random shapes, broadcast patterns, operand dtypes and memory layouts through the NumPy-in / NumPy-out path against the
oracle (NumPy semantics are the reference's: result shape = broadcast, dtype = promotion with Python scalars weak).

    python tools/shape_fuzz.py [--trials 300] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tests")]

FUNCS = [("potential_temperature", ("t", "p"), {}), ("relative_humidity_from_specific_humidity", ("t", "q", "p"), {}),
         ("saturation_vapour_pressure", ("t",), {}), ("dewpoint_from_specific_humidity", ("q", "p"), {}),
         ("ept_from_specific_humidity", ("t", "q", "p"), {"method": "ifs"}),
         ("wet_bulb_temperature_from_specific_humidity", ("t", "q", "p"), {"ept_method": "ifs", "t_method": "newton"}),
         ("lcl", ("t", "td", "p"), {"method": "davies"}), ("virtual_temperature", ("t", "q"), {}),
         ("wet_bulb_temperature_from_specific_humidity", ("t", "q", "p"), {"ept_method": "ifs", "t_method": "bisect"}),
         ("temperature_on_moist_adiabat", ("ept", "p"), {"ept_method": "bolton35", "t_method": "bisect"}),
         ("wet_bulb_potential_temperature_from_dewpoint", ("t", "td", "p"), {"ept_method": "bolton39", "t_method": "direct"}),
         ("saturation_specific_humidity_slope", ("t", "p"), {"phase": "ice"}),
         ("temperature_on_dry_adiabat", ("p", "t", "p"), {})]
RANGE = dict(t=(230.0, 310.0), td=(225.0, 300.0), q=(1e-5, 0.02), p=(2e4, 1.05e5), ept=(280.0, 360.0), r=(5.0, 100.0), tc=(-40.0, 35.0),
             w=(1e-5, 0.02), e=(10.0, 3000.0), es=(10.0, 4000.0), th=(250.0, 330.0), t2=(220.0, 300.0), p2=(1.5e4, 9e4))


def _all_cases():
    import _fuzz

    return [(f, tuple(k), kw) for f, k, kw in _fuzz._case_table()]


ALL = _all_cases()


def _scalar_kind(x):
    return "float" if type(x) is float else ("0-d array" if isinstance(x, np.ndarray) else type(x).__name__)


def operand(rng, key, full):
    lo, hi = RANGE[key]
    kind = rng.choice(["f32", "f64", "f64", "f32", "pyfloat", "list", "zerod"])  # (no integer arrays: the reference's es of an
    # int64 array is an int64 array of truncated values -- its result buffer is zeros_like(t) --, the library computes float64)
    nd = len(full)
    # broadcast pattern: drop leading dims, set some dims to 1
    keep_from = rng.integers(0, nd + 1)
    shape = [int(n) if rng.random() < 0.7 else 1 for n in full[keep_from:]]
    if kind == "pyfloat":
        return float(rng.uniform(lo, hi)), "pyfloat"
    if kind == "zerod":
        return np.array(rng.uniform(lo, hi), dtype=rng.choice([np.float32, np.float64])), "zerod"
    a = rng.uniform(lo, hi, shape)
    if kind == "int":
        if key == "q":
            kind = "f64"
        else:
            return np.round(a).astype(np.int64), "int"
    if kind == "list":
        return a.tolist(), "list"
    a = a.astype(np.float32 if kind == "f32" else np.float64)
    lay = rng.choice(["c", "t", "s", "neg"])
    if lay == "t" and a.ndim >= 2:
        a = np.ascontiguousarray(a.T).T  # same values, Fortran order
    elif lay == "s" and a.ndim >= 1 and a.shape[-1] > 1:
        big = np.repeat(a, 2, axis=-1)
        big[..., ::2] = a
        a = big[..., ::2]  # a strided view
    elif lay == "neg" and a.ndim >= 1:
        a = a[..., ::-1][..., ::-1].copy()[..., ::-1]  # negative stride
    return a, kind + ":" + lay


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reference", action="store_true", help="run the reference itself in place of the library (build container)")
    ap.add_argument("--all", action="store_true", help="draw from all 94 function x variant cases (tests/golden/_case_table.py)")
    ap.add_argument("--stream", action="store_true", help="send every NumPy call through the streamed path (slices of the leading "
                    "axis, uploader / downloader threads) by setting its threshold to one byte, with a small slice budget")
    ap.add_argument("--multi", type=int, default=0, help="inside ekm_hip.multi_gpu([0] * N): the sharded path on one GPU")
    ap.add_argument("--device", action="store_true", help="hand the array operands over as DeviceArrays (results come back as "
                    "DeviceArrays in the promotion's dtype: shapes and values are compared, not the reference's result typing)")
    a = ap.parse_args()
    from oracle import thermo_oracle as orc

    if a.reference:  # (build container only) the REFERENCE in place of the library: what the oracle gets wrong about conventions
        import types

        here = os.path.join(ROOT, "tests", "golden")
        sys.path[:0] = [here, os.path.join(here, "_standin"), os.path.join(os.environ.get("EKM_REFERENCE", "/root/reference"), "src")]
        from earthkit.meteo.thermo import array as ref

        ek = types.SimpleNamespace(thermo=ref)
    else:
        import ekm_hip as ek

    np.seterr(all="ignore")
    rng = np.random.default_rng(a.seed)
    bad = 0
    import contextlib

    ctx = contextlib.nullcontext()
    if not a.reference and (a.stream or a.multi):
        from ekm_hip import _engine

        _engine._STREAM_BYTES, _engine._TINY_BYTES = 1, 0
        if a.stream:
            os.environ["EKM_STREAM_BUDGET_BYTES"] = os.environ.get("EKM_STREAM_BUDGET_BYTES", str(24 << 20))
        if a.multi:
            ctx = ek.multi_gpu([0] * a.multi)
    with ctx:
        bad = _trials(a, rng, ek, orc)
    print(f"shape fuzz: {a.trials} trials, {bad} differences")
    return bad


def _trials(a, rng, ek, orc):
    bad = 0
    for trial in range(a.trials):
        func, keys, kw = (ALL if a.all else FUNCS)[rng.integers(len(ALL if a.all else FUNCS))]
        nd = int(rng.integers(1, 4))
        full = [int(rng.choice([1, 2, 3, 5, 8, 17, 64, 130])) for _ in range(nd)]
        ops, kinds = zip(*[operand(rng, k, full) for k in keys])
        try:
            want = getattr(orc, func)(*[np.copy(o) if isinstance(o, np.ndarray) else o for o in ops], **kw)
        except Exception:  # the reference raises (a list times a float, a mask index into a 0-d array): the library accepting
            continue       # such a call is a superset, like N-d input to the bisection; nothing to compare
        try:
            if a.device:
                dops = [ek.to_device(np.ascontiguousarray(o)) if isinstance(o, np.ndarray) and o.ndim else o for o in ops]
                if not any(isinstance(o, ek.DeviceArray) for o in dops):
                    continue
                got = getattr(ek.thermo, func)(*dops, **kw)
                got = tuple(g.to_host() for g in got) if isinstance(got, tuple) else got.to_host()
            else:
                got = getattr(ek.thermo, func)(*ops, **kw)
        except Exception as ex:
            print(f"trial {trial} {func} {kinds} shapes {[np.shape(o) for o in ops]}: library raises {type(ex).__name__}: {str(ex)[:120]}")
            bad += 1
            continue
        wl = want if isinstance(want, tuple) else (want,)
        gl = got if isinstance(got, tuple) else (got,)
        for k, (w, g) in enumerate(zip(wl, gl)):
            if not a.device and np.ndim(w) == 0 and _scalar_kind(w) != _scalar_kind(g):
                bad += 1
                print(f"trial {trial} {func}{kw}[{k}] {kinds}: a scalar result comes back as {_scalar_kind(g)}, the reference's is {_scalar_kind(w)}")
                continue
            w, g = np.asarray(w), np.asarray(g)
            # a float64 result of MIXED operands: the reference forms what depends on float32 operands alone in float32 (es of
            # a float32 t carries 2.5e-6), the library computes everything in float64 -- the comparison is float32-grade there
            mixed = any(k.startswith(("f32", "zerod")) for k in kinds)
            tol = 1e-4 if w.dtype == np.float32 else (1e-5 if mixed else 1e-7)
            if mixed and kw.get("t_method") == "newton" and w.dtype == np.float64:
                tol = 2e-3  # the reference's theta_e is float32 there, and bolton35's one Newton step amplifies 1e-7 of it a thousandfold
            if kw.get("t_method") == "bisect":
                tol = max(tol, 2.1 * (120.0 / 4096) / 230.0)  # a sign that is rounding noise moves the search by up to two quanta
            if a.device:  # the promotion's dtype (float32 unless an operand is float64): compare there
                w = w.astype(g.dtype)
                if func == "lcl" and k == 0 and w.shape != g.shape:
                    w = np.broadcast_to(w, g.shape)  # (a DeviceArray result has the full broadcast shape)
            ok = w.shape == g.shape and w.dtype == g.dtype
            if ok:
                both = np.isfinite(w) & np.isfinite(g)
                ok = np.array_equal(np.isnan(w), np.isnan(g)) and (not both.any() or float(np.max(np.abs(g[both] - w[both]) / np.maximum(np.abs(w[both]), 1e-30))) <= tol)
            if not ok:
                bad += 1
                print(f"trial {trial} {func}{kw}[{k}] {kinds} shapes {[np.shape(o) for o in ops]}: want {w.shape} {w.dtype}, got {g.shape} {g.dtype}"
                      + ("" if w.shape != g.shape else f" max rel {float(np.nanmax(np.abs(g.astype(np.float64) - w) / np.abs(w))):.2e}"))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)

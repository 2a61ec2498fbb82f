#!/usr/bin/env python3
"""Random surface-pressure shapes, dtypes, level subsets, output selections, vertical axes and layouts through the hybrid-level
functions against the oracle (`--reference`, build container: the reference itself in place of the library).

    python tools/shape_fuzz_vertical.py [--trials 200] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tests")]
OUT = ["full", "half", "delta", "alpha"]


def compare(what, got, want, tol, atol=0.0):
    got = got if isinstance(got, (tuple, list)) else (got,)
    want = want if isinstance(want, (tuple, list)) else (want,)
    if len(got) != len(want):
        print(what, "number of outputs", len(got), len(want))
        return 1
    bad = 0
    for k, (g, w) in enumerate(zip(got, want)):
        g, w = np.asarray(g), np.asarray(w)
        ok = g.shape == w.shape and g.dtype == w.dtype
        if ok:
            both = np.isfinite(w) & np.isfinite(g)
            ok = np.array_equal(np.isnan(w), np.isnan(g)) and (not both.any() or bool(np.all(np.abs(g[both].astype(np.float64) - w[both]) <= np.maximum(tol * np.abs(w[both]), atol))))
        if not ok:
            bad += 1
            print(f"{what}[{k}]: want {w.shape} {w.dtype}, got {g.shape} {g.dtype}"
                  + ("" if w.shape != g.shape else f" max rel {float(np.nanmax(np.abs(g.astype(np.float64) - w) / np.maximum(np.abs(w), 1.0))):.2e}"))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reference", action="store_true")
    a = ap.parse_args()
    from oracle import vertical_oracle as vo

    if a.reference:
        here = os.path.join(ROOT, "tests", "golden")
        sys.path[:0] = [here, os.path.join(here, "_standin"), os.path.join(os.environ.get("EKM_REFERENCE", "/root/reference"), "src")]
        from earthkit.meteo.vertical import array as lib
        import ekm_hip.vertical as ekv  # (only for the level tables)
        params = ekv.hybrid_level_parameters
    else:
        import ekm_hip as ek
        lib, params = ek.vertical, ek.vertical.hybrid_level_parameters
    np.seterr(all="ignore")
    rng = np.random.default_rng(a.seed)
    bad = 0
    for trial in range(a.trials):
        nlev = int(rng.choice([137, 91]))
        dtype = rng.choice([np.float32, np.float64])
        A, B = (x.astype(dtype) for x in params(nlev))
        shape = tuple(int(rng.choice([1, 2, 3, 7, 16, 33])) for _ in range(int(rng.integers(0, 4))))
        sp = rng.uniform(5e4, 1.08e5, shape).astype(dtype)
        if sp.ndim >= 2 and rng.random() < 0.3:
            sp = np.ascontiguousarray(sp.T).T
        tol = 1e-4 if dtype == np.float32 else 1e-7
        kind = rng.choice(["pressure", "pressure", "thickness", "geopotential", "height"])
        if kind == "pressure":
            levels = None
            if rng.random() < 0.5:
                n = int(rng.integers(1, 6))
                levels = sorted(int(x) for x in rng.choice(np.arange(1, nlev + 1), n, replace=False))
                if rng.random() < 0.3:
                    levels = levels[::-1]
            output = [str(x) for x in rng.choice(OUT, int(rng.integers(1, 5)), replace=False)]
            output = output[0] if len(output) == 1 and rng.random() < 0.5 else tuple(output)
            kw = dict(levels=levels, alpha_top=str(rng.choice(["ifs", "arpege"])), output=output)
            what = f"trial {trial} pressure_on_hybrid_levels nlev={nlev} {dtype.__name__} sp{shape} {kw}"
            try:
                want = vo.pressure_on_hybrid_levels(A, B, sp, **kw)
            except Exception:
                continue
            try:
                got = lib.pressure_on_hybrid_levels(A, B, sp, **kw)
            except Exception as ex:
                print(what, "raises", type(ex).__name__, str(ex)[:100])
                bad += 1
                continue
            # fp32: delta = log(p_lo/p_hi) and alpha = 1 - p/dp*delta are O(1) quantities formed from a ratio within 3e-3 of 1
            # at the lowest levels -- 6e-8 of the pressures is 2e-5 of delta and 1e-2 of alpha THERE in the reference's own fp32
            # run (its fp64 run: alpha[134] = 0.0014127, fp32: 0.0014289); the bar is absolute, 1e-4
            bad += compare(what, got, want, tol, 1e-4 if dtype == np.float32 else 0.0)
            continue
        # the chain: fields [nlev_used, *sp.shape] with the vertical axis moved somewhere
        used = nlev if rng.random() < 0.5 else int(rng.integers(2, nlev))  # the LOWEST `used` levels (vertical.py:1191-1203)
        base = (used,) + shape
        t = rng.uniform(200.0, 300.0, base).astype(dtype)
        q = rng.uniform(1e-6, 0.02, base).astype(dtype)
        zs = rng.uniform(0.0, 3e4, shape).astype(dtype)
        if shape and rng.random() < 0.25:  # one field constant along a column axis: [levels, .., 1, ..] broadcasts as in the reference
            ax = 1 + int(rng.integers(0, len(shape)))
            if rng.random() < 0.5:
                t = np.take(t, [0], axis=ax)
            else:
                q = np.take(q, [0], axis=ax)
        axis = int(rng.integers(0, len(base))) if rng.random() < 0.4 else 0
        if axis:
            t, q = np.moveaxis(t, 0, axis), np.moveaxis(q, 0, axis)
            if rng.random() < 0.5:
                t, q = np.ascontiguousarray(t), np.ascontiguousarray(q)
        at = str(rng.choice(["ifs", "arpege"]))
        if kind == "thickness":
            f, args, kw = "relative_geopotential_thickness_on_hybrid_levels", (t, q, A, B, sp), dict(alpha_top=at, vertical_axis=axis)
        elif kind == "geopotential":
            f, args, kw = "geopotential_on_hybrid_levels", (t, q, zs, A, B, sp), dict(alpha_top=at, vertical_axis=axis)
        else:
            f, args, kw = "height_on_hybrid_levels", (t, q, zs, A, B, sp), dict(alpha_top=at, vertical_axis=axis, h_type=str(rng.choice(["geometric", "geopotential"])),
                                                                               h_reference=str(rng.choice(["ground", "sea"])))
        what = f"trial {trial} {f} nlev={nlev} used={used} {dtype.__name__} sp{shape} t{t.shape}{'C' if t.flags.c_contiguous else 'v'} q{q.shape} {kw}"
        try:
            if axis:
                # the reference's chain with vertical_axis != 0 moves alpha and delta -- whose level axis IS 0 -- along with t
                # and q (vertical.py:960-966): it raises "operands could not be broadcast" unless the level count happens to
                # equal that column dimension, and then mixes the axes silently.  The library computes what the argument
                # means; it is compared with the level-major call moved back
                if a.reference:
                    continue
                kw0 = dict(kw, vertical_axis=0)
                want = np.moveaxis(getattr(vo, f)(np.moveaxis(args[0], axis, 0), np.moveaxis(args[1], axis, 0), *args[2:], **kw0), 0, axis)
            else:
                want = getattr(vo, f)(*args, **kw)
        except Exception:
            continue
        try:
            got = getattr(lib, f)(*args, **kw)
        except Exception as ex:
            print(what, "raises", type(ex).__name__, str(ex)[:100])
            bad += 1
            continue
        # fp32: the reference's own tolerance for its fp32 chain (its tests: atol 10 m2/s2 on geopotential; alpha = 1 - p/dp*
        # log(..) cancels in fp32) -- 10 m2/s2, i.e. 1 m on the heights
        atol = 0.0 if dtype == np.float64 else (1.05 if f == "height_on_hybrid_levels" else 10.0)
        bad += compare(what, got, want, tol, atol)
    print(f"vertical shape fuzz: {a.trials} trials, {bad} differences")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)

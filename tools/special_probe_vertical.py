#!/usr/bin/env python3
"""Special operands (NaN, infinities, zeros, negatives, 1e-30, 1e30) through the GPU's hybrid-level functions and
w_from_omega against the oracle: prints what differs (NaN / inf pattern, finite values beyond the bar).

    python tools/special_probe_vertical.py"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tests")]
import ekm_hip as ek  # noqa: E402
from oracle import vertical_oracle as vo  # noqa: E402
from oracle import wind_oracle as wo  # noqa: E402

np.seterr(all="ignore")
S = [np.nan, np.inf, -np.inf, 0.0, -0.0, -1.0, 1e-30, 1e30]
NLEV = 137


def compare(what, got, want, tol, cols, want64=None):
    """want64: the oracle in fp64 on the same (fp32) operands -- where the reference's own fp32 run is further than `tol` from
    it (alpha = 1 - p/dp*log(..) cancels in fp32: 1e-4 of a thin layer's thickness), the bar is four times that distance."""
    got = got if isinstance(got, tuple) else (got,)
    want = want if isinstance(want, tuple) else (want,)
    want64 = None if want64 is None else (want64 if isinstance(want64, tuple) else (want64,))
    n = 0
    for k, (g, w) in enumerate(zip(got, want)):
        g, w = np.asarray(g, np.float64), np.asarray(w, np.float64)
        assert g.shape == w.shape, (what, k, g.shape, w.shape)
        same = (g == w) | (np.isnan(g) & np.isnan(w))
        both = np.isfinite(g) & np.isfinite(w)
        rel = np.zeros_like(w)
        rel[both] = np.abs(g[both] - w[both]) / np.maximum(np.abs(w[both]), 1.0)
        bar = np.full(w.shape, tol)
        if want64 is not None:
            w64 = np.asarray(want64[k], np.float64)
            ok = both & np.isfinite(w64)
            bar[ok] = np.maximum(tol, 4.0 * np.abs(w[ok] - w64[ok]) / np.maximum(np.abs(w64[ok]), 1.0))
        bad = ~same & ~(both & (rel <= bar))
        n += int(bad.sum())
        for idx in np.argwhere(bad)[:6]:
            idx = tuple(idx)
            print(what, "output", k, "at", idx, "column", cols[idx[-1]] if cols is not None else "", "want", w[idx], "got", g[idx])
    print(f"{what}: {n} differences")
    return n


def main():
    total = 0
    for dtype, tol in ((np.float32, 1e-4), (np.float64, 1e-7)):
        tag = dtype.__name__
        A, B = (x.astype(dtype) for x in ek.vertical.hybrid_level_parameters(NLEV))
        sps = np.array(S + [101325.0, 5e4], dtype=dtype)
        for out in ("full", "half", "delta", "alpha", ("full", "half", "delta", "alpha")):
            for at in ("ifs", "arpege"):
                g = ek.vertical.pressure_on_hybrid_levels(A, B, sps, alpha_top=at, output=out)
                w = vo.pressure_on_hybrid_levels(A, B, sps, alpha_top=at, output=out)
                total += compare(f"{tag} pressure_on_hybrid_levels[{out},{at}]", g, w, tol, list(sps))
        vals = S + [None]
        cols = list(itertools.product(vals, repeat=4))  # (sp, t at level 70, q at level 100, zs)
        n = len(cols)
        sp = np.array([101325.0 if c[0] is None else c[0] for c in cols], dtype=dtype)
        zs = np.array([500.0 if c[3] is None else c[3] for c in cols], dtype=dtype)
        t = np.tile(np.linspace(220.0, 290.0, NLEV).astype(dtype)[:, None], (1, n))
        q = np.tile(np.linspace(1e-6, 0.01, NLEV).astype(dtype)[:, None], (1, n))
        for j, c in enumerate(cols):
            if c[1] is not None:
                t[70, j] = c[1]
            if c[2] is not None:
                q[100, j] = c[2]
        d64 = [x.astype(np.float64) for x in (t, q, zs, A, B, sp)] if dtype == np.float32 else None
        w64 = (lambda f, *a, **kw: None) if d64 is None else (lambda f, *a, **kw: f(*a, **kw))
        total += compare(f"{tag} relative_geopotential_thickness_on_hybrid_levels", ek.vertical.relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp),
                         vo.relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp), tol, cols,
                         w64(vo.relative_geopotential_thickness_on_hybrid_levels, *([d64[0], d64[1], d64[3], d64[4], d64[5]] if d64 else [])))
        total += compare(f"{tag} geopotential_on_hybrid_levels", ek.vertical.geopotential_on_hybrid_levels(t, q, zs, A, B, sp),
                         vo.geopotential_on_hybrid_levels(t, q, zs, A, B, sp), tol, cols, w64(vo.geopotential_on_hybrid_levels, *(d64 or [])))
        for ht in ("geometric", "geopotential"):
            for hr in ("ground", "sea"):
                total += compare(f"{tag} height_on_hybrid_levels[{ht},{hr}]", ek.vertical.height_on_hybrid_levels(t, q, zs, A, B, sp, h_type=ht, h_reference=hr),
                                 vo.height_on_hybrid_levels(t, q, zs, A, B, sp, h_type=ht, h_reference=hr), tol, cols,
                                 w64(vo.height_on_hybrid_levels, *(d64 or []), h_type=ht, h_reference=hr))
        c3 = np.array(list(itertools.product(S + [0.5], S + [280.0], S + [9e4])), dtype=dtype)
        total += compare(f"{tag} w_from_omega", ek.wind.w_from_omega(c3[:, 0].copy(), c3[:, 1].copy(), c3[:, 2].copy()),
                         wo.w_from_omega(c3[:, 0].copy(), c3[:, 1].copy(), c3[:, 2].copy()), tol, [tuple(r) for r in c3])
    print("total differences", total)
    return total


if __name__ == "__main__":
    main()

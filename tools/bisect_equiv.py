#!/usr/bin/env python3
"""The fp32 IFS bisection as a tree walk with transcendental-free sign tests (csrc/thermo_math.hpp::t_on_ma_bisect_heap)
against (a) itself with the reference's residual evaluated at EVERY step (tuning parameter bisect_exact) and (b) the
round-3 library's stepwise search (earthkit-meteo_amd/variants/r03/libekm_thermo.so, when present): every point of the
3600 x 1800 x 137 benchmark field, p as a field / level vector / hybrid levels, bit for bit.

    python tools/bisect_equiv.py [--levels 137] [--chunk 8]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tools")]
from ekm_hip import _ffi  # noqa: E402
from sweep import load  # noqa: E402

INNER = 1800 * 3600


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"],
                    help="f64: the fp64 walk (fp32 sign tests, fp64 residual on ambiguous steps) against the same walk with the fp64 "
                         "residual at every step; the round-3 library is then compared in quanta (its fp64 exp2 differs by 1e-12)")
    ap.add_argument("--method", default="ifs", choices=["ifs", "bolton35", "bolton39"])
    a = ap.parse_args()
    meth = {"ifs": 0, "bolton35": 1, "bolton39": 2}[a.method]
    tag, npdt, isz = a.dtype, (np.float32 if a.dtype == "f32" else np.float64), (4 if a.dtype == "f32" else 8)
    new = load(_ffi.library_path())
    old_path = os.path.join(ROOT, "earthkit-meteo_amd", "variants", "r03", "libekm_thermo.so")
    old = load(old_path) if os.path.exists(old_path) else None
    chk = lambda rc: rc >= 0 or sys.exit(f"error {rc}: {new.ekm_last_error().decode()}")  # noqa: E731
    dev, nmax = 0, a.chunk * INNER

    def dmalloc(nbytes):
        p = C.c_void_p()
        chk(new.ekm_malloc(dev, nbytes, C.byref(p)))
        return p.value

    t, q, p, o1, o2, o3 = (dmalloc(isz * nmax) for _ in range(6))
    plev = dmalloc(isz * 137)
    chk(getattr(new, f'ekm_synth_levels_{tag}')(dev, None, plev, a.levels))
    from ekm_hip.vertical import hybrid_level_parameters

    A, B = (x[137 - a.levels:].astype(npdt) for x in hybrid_level_parameters(137))
    sp = (101325.0 * (1.0 - 0.35 * np.random.default_rng(1).random(INNER) ** 3)).astype(npdt)
    dA, dB, dsp = dmalloc(A.nbytes), dmalloc(B.nbytes), dmalloc(sp.nbytes)
    for d, h in ((dA, A), (dB, B), (dsp, sp)):
        chk(new.ekm_h2d(dev, d, h.ctypes.data, h.nbytes, None))
    F = _ffi.Operand
    wb_new = getattr(new, f'ekm_wet_bulb_temperature_from_specific_humidity_{tag}')
    wb_old = getattr(old, f'ekm_wet_bulb_temperature_from_specific_humidity_{tag}') if old is not None else None
    host = [np.empty(nmax, npdt) for _ in range(3)]
    total = {m: [0, 0, 0] for m in ("field", "level", "hybrid")}  # points, default != all-exact, default != r03
    for lo in range(0, a.levels, a.chunk):
        hi = min(a.levels, lo + a.chunk)
        n = (hi - lo) * INNER
        for mode in ("field", "level", "hybrid"):
            if mode == "hybrid":
                ptmp = p
                chk(getattr(new, f'ekm_pressure_on_hybrid_levels_{tag}')(dev, None, dA + isz * lo, dB + isz * lo, dsp, INNER, hi - lo, None, None,
                                                          int(lo == 0 and A[0] == 0 and B[0] == 0), float(np.log(2)), ptmp, None, None, None))
                chk(getattr(new, f'ekm_synth_fill_given_p_{tag}')(dev, None, t, q, ptmp, lo * INNER, n, 20260313))
                nz = np.flatnonzero(B[lo:hi + 1] != 0.0)
                nflat = int(max(0, (nz[0] if nz.size else hi + 1 - lo) - 1))
                op_p = F(dsp, _ffi.HYBRID_FULL, nflat, hi - lo, INNER, dA + isz * lo, dB + isz * lo)
            else:
                chk(getattr(new, f'ekm_synth_fill_{tag}')(dev, None, t, q, p, lo * INNER, n, INNER, a.levels, 20260313))
                op_p = F(p, _ffi.FIELD, 0, 0, 0) if mode == "field" else F(plev + isz * lo, _ffi.LEVEL_MAJOR, 0, hi - lo, INNER)
            ops = [C.byref(F(t, _ffi.FIELD, 0, 0, 0)), C.byref(F(q, _ffi.FIELD, 0, 0, 0)), C.byref(op_p)]
            chk(new.ekm_set_tuning_param(b"bisect_exact", 0))
            chk(wb_new(dev, None, *ops, meth, 0, o1, n))
            chk(new.ekm_set_tuning_param(b"bisect_exact", 1))
            chk(wb_new(dev, None, *ops, meth, 0, o2, n))
            chk(new.ekm_set_tuning_param(b"bisect_exact", 0))
            if old is not None:
                chk(wb_old(dev, None, *ops, meth, 0, o3, n))
            chk(new.ekm_sync(dev))
            for d, h in zip((o1, o2, o3), host):
                chk(new.ekm_d2h(dev, h.ctypes.data, d, isz * n, None))
            chk(new.ekm_sync(dev))
            u1, u2, u3 = (h[:n].view(np.uint32 if isz == 4 else np.uint64) for h in host)
            nan1, nan2, nan3 = (np.isnan(h[:n]) for h in host)
            total[mode][0] += n
            total[mode][1] += int((~((u1 == u2) | (nan1 & nan2))).sum())
            if old is not None:
                if isz == 4:
                    total[mode][2] += int((~((u1 == u3) | (nan1 & nan3))).sum())
                else:  # the round-3 fp64 primitives differ by ~1e-12: a sign can flip only where its residual is that small
                    with np.errstate(all="ignore"):
                        far = np.abs(host[0][:n] - host[2][:n]) > 2.0001 * 120.0 / 4096.0
                    total[mode][2] += int((far | (nan1 != nan3)).sum())
        print(f"levels {lo}..{hi - 1}: " + ", ".join(f"{m} {v[1]}/{v[2]}" for m, v in total.items()), flush=True)
    for m, (n, d2, d3) in total.items():
        print(f"{m:6s}: {n} points; tree walk vs the same walk with the exact residual at every step: {d2} differ; "
              f"vs the round-3 stepwise search: {d3 if old is not None else 'n/a'} differ")
    sys.exit(1 if any(v[1] or v[2] for v in total.values()) else 0)


if __name__ == "__main__":
    main()

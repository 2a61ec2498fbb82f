#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants and launch tunings on one GPU, one process
(cdna_hip_programming.md rule 24): for every (library variant, tiles, unroll,
workload) run R rounds of K launches, report median/min kernel ms and GB/s.

    python tools/sweep.py --libs default,variants/nont/libekm_thermo.so --workloads p3,full \
        --tiles 4,8,16 --unroll 1,2 --rounds 5 --steps 5 --levels 137
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
from ekm_hip import _ffi  # noqa: E402

INNER = int(os.environ.get("EKM_SWEEP_INNER", str(1800 * 3600)))  # points per level
W = {  # workload: entry, operands, ints, nout, bytes/pt (field p)
    "full": ("pipeline_full", "tqp", (), 6, 36), "p3": ("pipeline_svp_td_rh", "tqp", (), 3, 24),
    "wetbulb": ("wet_bulb_temperature_from_specific_humidity", "tqp", (0, 1), 1, 16),
    "wetbulb_bisect": ("wet_bulb_temperature_from_specific_humidity", "tqp", (0, 0), 1, 16),
    "wetbulb_bisect_bolton35": ("wet_bulb_temperature_from_specific_humidity", "tqp", (1, 0), 1, 16),
    "wetbulb_bisect_bolton39": ("wet_bulb_temperature_from_specific_humidity", "tqp", (2, 0), 1, 16),
    "wetbulb_td": ("wet_bulb_temperature_from_dewpoint", "tqp", (0, 1), 1, 16),
    "wetbulb_bolton35": ("wet_bulb_temperature_from_specific_humidity", "tqp", (1, 1), 1, 16),
    "wetbulb_bolton39": ("wet_bulb_temperature_from_specific_humidity", "tqp", (2, 1), 1, 16),
    "t_on_ma_bisect": ("temperature_on_moist_adiabat", "tp", (0, 0), 1, 12),
    "rh": ("relative_humidity_from_specific_humidity", "tqp", (), 1, 16),
    "ept": ("ept_from_specific_humidity", "tqp", (0,), 1, 16),
    "theta": ("potential_temperature", "tp", (), 1, 12), "svp": ("saturation_vapour_pressure", "t", (0,), 1, 8),
    "td": ("dewpoint_from_specific_humidity", "qp", (), 1, 12), "c2k": ("celsius_to_kelvin", "t", (), 1, 8),
}


def load(path):
    lib = C.CDLL(path)
    for name, (args, res) in _ffi._signatures().items():
        fn = getattr(lib, name, None)  # an older variant library may lack the newest entry points
        if fn is not None:
            fn.argtypes, fn.restype = args, res
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--workloads", default="p3,full")
    ap.add_argument("--tiles", default="8")
    ap.add_argument("--unroll", default="1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--pmode", default="field")
    ap.add_argument("--out", default="")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--nflat", type=int, default=-1, help="hybrid: leading pure pressure levels passed to the library (-1 = count them)")
    ap.add_argument("--skew", default="0", help="comma list: byte offset added to the k-th array's base (k*skew)")
    ap.add_argument("--smooth", action="store_true",
                    help="overwrite the generator's fields (N(0, 8 K) and U(1, 100 %) PER POINT: neighbouring lanes hundreds of search-tree "
                         "leaves apart) with horizontally smooth ones of the same per-level ranges: T = T_std(p_k) + 8 K x a low-wavenumber "
                         "pattern, RH and p likewise -- what an analysis field looks like to a wave (VERDICT r5 item 2d); fp32, p a field or levels")
    ap.add_argument("--params", default="", help="secondary parameters to sweep, e.g. hybrid_band_kb=512:4096:32768,"
                                                   "lev_per_wg=1:2 (ekm_set_tuning_param); the cross product is run")
    a = ap.parse_args()

    libs = {}
    for spec in a.libs.split(","):
        path = _ffi.library_path() if spec == "default" else os.path.join(ROOT, "earthkit-meteo_amd", spec)
        libs[spec] = load(path)
    base = next(iter(libs.values()))
    chk = lambda rc: rc >= 0 or sys.exit(f"error {rc}: {base.ekm_last_error().decode()}")  # noqa: E731
    dev, n = 0, a.levels * INNER
    isz = 4 if a.dtype == "f32" else 8

    skews = [int(x) for x in a.skew.split(",")]
    maxskew = max(skews) * 10

    def dmalloc(nbytes):
        p = C.c_void_p()
        chk(base.ekm_malloc(dev, nbytes + maxskew, C.byref(p)))
        return p.value

    bases = [dmalloc(isz * n) for _ in range(9)]
    pl = dmalloc(isz * a.levels)
    t, q, p = bases[:3]
    outs = bases[3:]
    chk(getattr(base, f'ekm_synth_fill_{a.dtype}')(dev, None, t, q, p, 0, n, INNER, a.levels, 20260313))
    chk(getattr(base, f'ekm_synth_levels_{a.dtype}')(dev, None, pl, a.levels))
    if a.smooth:
        import numpy as np

        from oracle import synthetic
        from oracle import thermo_oracle as orc

        dt = np.float32 if a.dtype == "f32" else np.float64
        nlat = 1800 if INNER == 1800 * 3600 else 1
        y, x = np.meshgrid(np.linspace(0, 1, nlat, endpoint=False), np.linspace(0, 1, INNER // nlat, endpoint=False), indexing="ij")
        pat = [(np.sin(2 * np.pi * (3 * x + k1 * y)) * np.cos(2 * np.pi * (2 * y + k2 * x))).ravel() for k1, k2 in ((1, 0.5), (2, 1.5), (0.5, 1))]
        pl_host = synthetic.level_pressures(137)[np.linspace(0, 136, a.levels).round().astype(int)] if a.levels != 137 else synthetic.level_pressures(137)
        for k in range(a.levels):
            pk = pl_host[k] * (1.0 + 0.05 * pat[2]) if a.pmode == "field" else np.full(INNER, pl_host[k])
            tk = np.clip(synthetic.standard_temperature(pk) + 8.0 * pat[0], 180.0, 330.0)
            with np.errstate(all="ignore"):
                qk = orc.specific_humidity_from_relative_humidity(tk, 50.5 + 49.5 * pat[1], pk)
            qk = np.where(np.isnan(qk), 3e-6, np.minimum(qk, 0.04))
            for ptr, arr in ((t, tk), (q, qk), (p, pk)):
                h = np.ascontiguousarray(arr.astype(dt))
                chk(base.ekm_h2d(dev, ptr + k * INNER * isz, h.ctypes.data, h.nbytes, None))
            chk(base.ekm_sync(dev))
        print(f"# --smooth: t, q, p overwritten with low-wavenumber fields ({a.levels} levels)", flush=True)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    chk(base.ekm_event_create(dev, C.byref(ev0)))
    chk(base.ekm_event_create(dev, C.byref(ev1)))
    F = _ffi.Operand
    hyb = None
    if a.pmode == "hybrid" or "geopotential" in a.workloads:
        import numpy as np

        from ekm_hip.vertical import hybrid_level_parameters

        dt = np.float32 if a.dtype == "f32" else np.float64
        A, B = (x[137 - a.levels:].astype(dt) for x in hybrid_level_parameters(137))
        sp = (101325.0 * (1.0 - 0.35 * np.random.default_rng(1).random(INNER) ** 3)).astype(dt)
        zs = np.maximum(0.0, (101325.0 - sp.astype(np.float64)) / 1.2).astype(dt)
        hyb = []
        for arr in (A, B, sp, zs):
            ptr = dmalloc(arr.nbytes)
            chk(base.ekm_h2d(dev, ptr, arr.ctypes.data, arr.nbytes, None))
            hyb.append(ptr)
        chk(base.ekm_sync(dev))
    ops = {"t": F(t, 0, 0, 0, 0), "q": F(q, 0, 0, 0, 0),
           "p": F(p, 0, 0, 0, 0) if a.pmode == "field" else (
               F(pl, 2, 0, a.levels, INNER) if a.pmode == "level" else F(hyb[2], 4, a.nflat if a.nflat >= 0 else int(max(0, np.flatnonzero(B != 0)[0] - 1)), a.levels, INNER, hyb[0], hyb[1]))}
    W["geopotential"] = ("geopotential_on_hybrid_levels", "", (), 1, 12)
    pnames = [kv.split("=")[0] for kv in a.params.split(",") if kv]
    pvals = [[int(v) for v in kv.split("=")[1].split(":")] for kv in a.params.split(",") if kv]
    import itertools

    psets = list(itertools.product(*pvals)) if pnames else [()]

    configs = [(ln, int(b), int(u), w, sk, ps) for ln in libs for b in a.tiles.split(",") for u in a.unroll.split(",")
               for w in a.workloads.split(",") for sk in skews for ps in psets]
    times = {c: [] for c in configs}
    for rnd in range(a.rounds + 1):  # round 0 = warm-up
        for c in configs:
            ln, b, u, w, sk, ps = c
            lib = libs[ln]
            entry, which, ints, nout, bpp = W[w]
            chk(lib.ekm_set_tuning(b, u))
            for pn, pv in zip(pnames, ps):
                chk(lib.ekm_set_tuning_param(pn.encode(), pv))
            fn = getattr(lib, f"ekm_{entry}_{a.dtype}")
            sops = {"t": F(t, 0, 0, 0, 0), "q": F(q + sk, 0, 0, 0, 0),
                    "p": F(p + 2 * sk, 0, 0, 0, 0) if a.pmode == "field" else ops["p"]}
            cargs = [dev, None] + [C.byref(sops[k]) for k in which] + list(ints) + \
                [o + (3 + i) * sk for i, o in enumerate(outs[:nout])] + [n]
            if w == "geopotential":
                cargs = [dev, None, hyb[0], hyb[1], hyb[2], hyb[3], t, q, INNER, a.levels, 1, 0.6931471805599453, 1, outs[0]]
            chk(fn(*cargs))
            chk(base.ekm_event_record(dev, ev0, None))
            for _ in range(a.steps):
                chk(fn(*cargs))
            chk(base.ekm_event_record(dev, ev1, None))
            chk(base.ekm_sync(dev))
            ms = C.c_float()
            chk(base.ekm_event_elapsed_ms(dev, ev0, ev1, C.byref(ms)))
            if rnd:
                times[c].append(ms.value / a.steps)
    rows = []
    for c in configs:
        ln, b, u, w, sk, ps = c
        bpp = (W[w][4] - (4 if a.pmode in ("level", "hybrid") and "p" in W[w][1] else 0)) * isz // 4
        med, mn = statistics.median(times[c]), min(times[c])
        rows.append(dict(lib=ln, tiles=b, unroll=u, workload=w, skew=sk, params=dict(zip(pnames, ps)), med_ms=round(med, 4), min_ms=round(mn, 4),
                         gbs_med=round(bpp * n / med / 1e6, 1), frac=round(bpp * n / med / 1e6 / 8000, 4)))
        print(f"{ln:28s} tiles={b:<5d} u={u} skew={sk:<8d} {dict(zip(pnames, ps))!s:36s} {w:15s} med {med:8.4f} ms  min {mn:8.4f} ms  {rows[-1]['gbs_med']:8.1f} GB/s"
              f"  {rows[-1]['frac'] * 100:5.1f}%", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The wide-domain fuzz of tests/_fuzz.py with other seeds and more points than the suites run (they take one seed of
2^20 points): every theta_e method x {bisect, newton} x {fp32, fp64} x {from (theta_e, p), from (t, q, p), from (t, td, p), theta_w from both} on the GPU
against the oracle, judged by the same rules (tests/_fuzz.py::judge raises on the first real miss), plus the default walk
against the exact walk bit for bit.

    python tools/fuzz_sweep.py [--seeds 3] [--n 4194304]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd"), os.path.join(ROOT, "tests")]

import _fuzz  # noqa: E402


def twin_sweep(a):
    """The same sweep on the host twin (the kernels' templates compiled for the CPU; no GPU): --twin."""
    import _hosttwin as twin

    t0 = time.time()
    for s in range(a.seeds):
        seed = a.first_seed + 17 * s
        for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
            d = _fuzz.make(a.n, seed, dtype, a.adversarial)
            for func, keys, method, tm in _fuzz.CASES + _fuzz.CASES_MORE:
                ins, kw = [d[k] for k in keys], dict(ept_method=method, t_method=tm)
                os.environ.pop("EKM_TWIN_BISECT_EXACT", None)
                got = twin.by_reference_name(func, ins, kw, dtype)
                line = _fuzz.judge(func, keys, method, tm, tag, d, got, limits=a.adversarial != 2, sweep=True, min_identical=0.99 if a.adversarial else 0.999)
                if tm == "bisect":
                    os.environ["EKM_TWIN_BISECT_EXACT"] = "1"
                    e = twin.by_reference_name(func, ins, kw, dtype)
                    os.environ.pop("EKM_TWIN_BISECT_EXACT", None)
                    diff = ~((got == e) | (np.isnan(got) & np.isnan(e)))
                    assert not diff.any(), (func, method, tag, seed, int(diff.sum()), np.flatnonzero(diff)[:4], got[diff][:4], e[diff][:4])
                    line += "; default walk == exact walk on every point"
                print(f"seed {seed} {line}", flush=True)
    print(f"fuzz sweep (host twin): {a.seeds} seeds x {a.n} points x {2 * len(_fuzz.CASES + _fuzz.CASES_MORE)} cases: no real miss, {time.time() - t0:.0f} s")


def ill_conditioned_sweep(a):
    """README "Parity" names two regions it excepts; this prints, per seed, how often the kernels (or the host twin) miss the
    fp32 reference there against how often the reference's own fp32 run misses its fp64 run."""
    if a.twin:
        import _hosttwin as twin

        run = lambda func, ins, kw: twin.by_reference_name(func, ins, dict(kw), np.float32)  # noqa: E731
    else:
        import ekm_hip

        run = lambda func, ins, kw: getattr(ekm_hip.thermo, func)(*ins, **kw)  # noqa: E731
    for s in range(a.seeds):
        seed = a.first_seed + 17 * s
        for kind in sorted(_fuzz.ILL_CONDITIONED):
            d = _fuzz.make_ill_conditioned(kind, n=a.n, seed=seed)
            func, keys, kw = _fuzz.ILL_CONDITIONED[kind]
            print(f"seed {seed} {_fuzz.judge_vs_reference_spread(kind, d, run(func, [d[k] for k in keys], kw))}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--n", type=int, default=1 << 22)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--adversarial", type=int, default=0, nargs="?", const=1, help="tests/_fuzz.py::make(adversarial=N): 1 = points moved "
                    "next to the search tree's node pressures, to p0 and to saturation; 2 = also theta_e next to the Davies-Jones "
                    "regime thresholds (then without the limits on how many points may need an explanation)")
    ap.add_argument("--twin", action="store_true", help="run the host twin instead of the GPU")
    ap.add_argument("--ill-conditioned", action="store_true", help="the two regions of physical input where the reference's own fp32 and "
                    "fp64 Newton runs disagree beyond 1e-4 (tests/_fuzz.py::ILL_CONDITIONED): ours against the reference's own spread")
    a = ap.parse_args()
    np.seterr(all="ignore")
    if a.ill_conditioned:
        return ill_conditioned_sweep(a)
    if a.twin:
        return twin_sweep(a)
    import ekm_hip
    from ekm_hip import _ffi

    lib = _ffi.lib()
    t0 = time.time()
    for s in range(a.seeds):
        seed = a.first_seed + 17 * s
        for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
            d = _fuzz.make(a.n, seed, dtype, a.adversarial)
            dd = {k: ekm_hip.to_device(v) for k, v in d.items()}
            for func, keys, method, tm in _fuzz.CASES + _fuzz.CASES_MORE:
                ins = [dd[k] for k in keys]
                out = getattr(ekm_hip.thermo, func)(*ins, ept_method=method, t_method=tm)
                got = out.to_host()
                out.free()
                line = _fuzz.judge(func, keys, method, tm, tag, d, got, limits=a.adversarial != 2, sweep=True, min_identical=0.99 if a.adversarial else 0.999)
                if tm == "bisect":
                    _ffi.check(lib.ekm_set_tuning_param(b"bisect_exact", 1))
                    ex = getattr(ekm_hip.thermo, func)(*ins, ept_method=method, t_method=tm)
                    _ffi.check(lib.ekm_set_tuning_param(b"bisect_exact", 0))
                    e = ex.to_host()
                    ex.free()
                    diff = int((~((got == e) | (np.isnan(got) & np.isnan(e)))).sum())
                    assert diff == 0, (func, method, tag, seed, diff)
                    line += "; default walk == exact walk on every point"
                print(f"seed {seed} {line}", flush=True)
            for v in dd.values():
                v.free()
    print(f"fuzz sweep: {a.seeds} seeds x {a.n} points x {2 * len(_fuzz.CASES + _fuzz.CASES_MORE)} cases: no real miss, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Run the five configurations BASELINE.json names, one after the other, on one GPU, with
parity against the oracle and the NumPy oracle timed beside each (1 core).

    python tools/bench_configs.py [--out gpurun_out/configs.json]

cfg1  thermo.potential_temperature on 2-element NumPy arrays (README quick-start; plumbing)
cfg2  thermo.relative_humidity_from_specific_humidity on 721x1440 fp64, 1 GPU vs NumPy
cfg3  fused svp -> dewpoint -> rh pipeline on 3600x1800x137 fp32 (kernel, HBM-resident)
cfg4  wet-bulb temperature (Newton) on 3600x1800x137 fp32
cfg5  full thermo pipeline on 3600x1800x137 fp32 (the single-GPU shard of the scaling curve)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]

import ekm_hip  # noqa: E402
from ekm_hip import _ffi, thermo  # noqa: E402
from oracle import synthetic  # noqa: E402
from oracle import thermo_oracle as orc  # noqa: E402

np.seterr(all="ignore")


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def maxrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    r = np.abs(a - b) / np.abs(b)
    return float(np.nanmax(r)), int((np.isnan(a) != np.isnan(b)).sum())


def kernel_ms(fn, reps=10):
    lib = _ffi.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    _ffi.check(lib.ekm_event_create(0, C.byref(e0)))
    _ffi.check(lib.ekm_event_create(0, C.byref(e1)))
    fn()
    ekm_hip.synchronize()
    _ffi.check(lib.ekm_event_record(0, e0, None))
    for _ in range(reps):
        fn()
    _ffi.check(lib.ekm_event_record(0, e1, None))
    ekm_hip.synchronize()
    ms = C.c_float()
    _ffi.check(lib.ekm_event_elapsed_ms(0, e0, e1, C.byref(ms)))
    return ms.value / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = {}

    # cfg1 -------------------------------------------------------------------------------------
    t = np.array([264.12, 261.45])
    p = np.array([85000.0, 85000.0])
    got = thermo.potential_temperature(t, p)
    res["cfg1"] = {"what": "potential_temperature, README vector, fp64, NumPy in/out",
                   "result": got.tolist(), "expected": [276.672291, 273.87539937],
                   "max_rel_err": maxrel(got, orc.potential_temperature(t, p))[0],
                   "call_latency_ms": best(lambda: thermo.potential_temperature(t, p), 20) * 1e3,
                   "numpy_latency_ms": best(lambda: orc.potential_temperature(t, p), 20) * 1e3}

    # cfg2 -------------------------------------------------------------------------------------
    n2 = 721 * 1440
    t, q, p, _ = synthetic.make_fields(1, n2, dtype=np.float64, levels=[114])  # p_k ~ 850 hPa, field mode
    t, q, p = (x.reshape(721, 1440) for x in (t, q, p))
    want = orc.relative_humidity_from_specific_humidity(t, q, p)
    got = thermo.relative_humidity_from_specific_humidity(t, q, p)
    dt, dq, dp = (ekm_hip.to_device(x) for x in (t, q, p))
    kms = kernel_ms(lambda: thermo.relative_humidity_from_specific_humidity(dt, dq, dp).free(), 20)
    e2e = best(lambda: thermo.relative_humidity_from_specific_humidity(t, q, p), 10)
    cpu = best(lambda: orc.relative_humidity_from_specific_humidity(t, q, p), 5)
    err, nanmm = maxrel(got, want)
    res["cfg2"] = {"what": "relative_humidity_from_specific_humidity, 721x1440 fp64", "points": n2,
                   "max_rel_err": err, "nan_mismatch": nanmm, "tolerance": 1e-6,
                   "numpy_1core_ms": cpu * 1e3, "gpu_numpy_in_out_ms": e2e * 1e3,
                   "gpu_device_resident_ms_per_call": kms,
                   "note": "33 MB total: fits the 256 MiB Infinity Cache; latency/launch-bound, not a bandwidth claim; "
                           "NumPy in/out is dominated by PCIe + allocation"}

    # cfg3-5 -----------------------------------------------------------------------------------
    nlev, inner = 137, 1800 * 3600
    n = nlev * inner
    lib = _ffi.lib()
    dev = [ekm_hip.DeviceArray.empty((nlev, inner), np.float32) for _ in range(3)]
    _ffi.check(lib.ekm_synth_fill_f32(0, None, dev[0].ptr, dev[1].ptr, dev[2].ptr, 0, n, inner, nlev, 20260313))
    outs = [ekm_hip.DeviceArray.empty((n,), np.float32) for _ in range(6)]
    F = _ffi.Operand
    ops = [C.byref(F(d.ptr, 0, 0, 0, 0)) for d in dev]

    def sample(arrs):  # 256-point windows from 32 levels
        idx = [int(l) * inner + 4321 for l in np.linspace(0, nlev - 1, 32).round()]
        return [np.concatenate([x.ravel().flat_slice(i, i + 256).to_host() for i in idx]) for x in arrs]

    hin = sample(dev)
    cfgs = {
        "cfg3": ("pipeline_svp_td_rh", 3, (), 24, lambda: orc.pipeline_svp_td_rh(*hin)),
        "cfg4": ("wet_bulb_temperature_from_specific_humidity", 1, (0, 1), 16,
                 lambda: (orc.wet_bulb_temperature_from_specific_humidity(*hin, "ifs", "newton"),)),
        "cfg5": ("pipeline_full", 6, (), 36, lambda: orc.pipeline_full(*hin)),
    }
    # CPU: the oracle on one level slab (6.48 M points), 1 core
    ht = [x.ravel().flat_slice(100 * inner, 101 * inner).to_host() for x in dev]
    for key, (entry, nout, ints, bpp, oracle) in cfgs.items():
        fn = getattr(lib, f"ekm_{entry}_f32")
        cargs = [0, None] + ops + list(ints) + [o.ptr for o in outs[:nout]] + [n]
        ms = kernel_ms(lambda: _ffi.check(fn(*cargs)), 10)
        hout = sample(outs[:nout])
        errs = [maxrel(g, w) for g, w in zip(hout, oracle())]
        if key == "cfg3":
            cpu = best(lambda: orc.pipeline_svp_td_rh(*ht), 2)
        elif key == "cfg4":
            cpu = best(lambda: orc.wet_bulb_temperature_from_specific_humidity(*ht, "ifs", "newton"), 2)
        else:
            cpu = best(lambda: orc.pipeline_full(*ht), 2)
        res[key] = {"what": f"{entry} on 137x1800x3600 fp32, HBM-resident", "points": n, "kernel_ms": ms,
                    "grid_points_per_s": n / ms * 1e3, "bytes_per_point": bpp, "GBps": bpp * n / ms / 1e6,
                    "frac_of_8TBps": bpp * n / ms / 1e6 / 8000, "max_rel_err": max(e[0] for e in errs),
                    "nan_mismatch": sum(e[1] for e in errs), "tolerance": 1e-4,
                    "numpy_1core_grid_points_per_s": inner / cpu}
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

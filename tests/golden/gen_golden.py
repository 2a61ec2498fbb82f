#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE itself.

Run in the build container only (the reference does not travel to the GPU box):

    python tests/golden/gen_golden.py

It imports ``earthkit.meteo.thermo`` from /root/reference/src with the stand-in
for the un-vendored ``earthkit.utils.array`` (tests/golden/_standin), evaluates
every public thermo function on several input sets in fp64 and fp32 and writes

    tests/golden/thermo_golden.npz   inputs + reference outputs (+ JSON manifest)
    tests/golden/ref_csv.npz         the data of the reference's own CSV fixtures
                                     (tests/data/*.csv there), column by column

Only data (inputs and expected outputs) is written; no reference source.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("EKM_REFERENCE", "/root/reference")
sys.path[:0] = [HERE, os.path.join(HERE, "_standin"), os.path.join(REF, "src"), ROOT]

from earthkit.meteo.thermo import array as ref  # noqa: E402

from oracle import synthetic  # noqa: E402

warnings.simplefilter("ignore")
np.seterr(all="ignore")

PHASES = ["mixed", "water", "ice"]
EPT = ["ifs", "bolton35", "bolton39"]


def read_csv(name):
    return np.genfromtxt(os.path.join(REF, "tests", "data", name), delimiter=",", names=True)


def derived_pool(base, dtype):
    """Inputs every function draws its arguments from (all 1-D, same dtype)."""
    t, td, r, q, p = (np.asarray(base[k], dtype=dtype) for k in ("t", "td", "r", "q", "p"))
    pool = dict(t=t, td=td, r=r, q=q, p=p)
    pool["tc"] = (t - dtype(273.16)).astype(dtype)
    pool["w"] = (q / (1 - q)).astype(dtype)
    pool["e"] = ref.vapour_pressure_from_specific_humidity(q, p).astype(dtype)
    pool["es"] = ref.saturation_vapour_pressure(t).astype(dtype)
    pool["th"] = ref.potential_temperature(t, p).astype(dtype)
    pool["ept"] = ref.ept_from_specific_humidity(t, q, p).astype(dtype)
    pool["t2"] = (t - dtype(10.0)).astype(dtype)
    pool["p2"] = (p * dtype(0.8)).astype(dtype)
    return pool


from _case_table import case_table  # noqa: E402


def edge_base():
    """Inputs that force the NaN / inf / threshold paths."""
    TI = 273.16 - 23
    t = [TI, np.nextafter(TI, 0), np.nextafter(TI, 1e9), 273.16, np.nextafter(273.16, 0), np.nextafter(273.16, 1e9),
         261.0, 300.0, 330.0, 180.0, 372.0, 400.0, 290.0, 290.0, 290.0, 290.0, 250.0, 305.0, 323.0, 310.0,
         np.nan, 285.0, 285.0, 32.19, 20.0, np.inf]
    n = len(t)
    p = np.full(n, 90000.0)
    q = np.full(n, 0.005)
    r = np.full(n, 60.0)
    td = np.array(t) - 4.0
    # saturated / super-saturated air at low pressure: p - es < eps, tw <= 0 branches
    p[10], p[11] = 90000.0, 50000.0      # es(372 K) ~ 0.96e5 > p, es(400 K) >> p
    q[12], q[13], q[14], q[15] = 0.0, -0.001, 0.05, 0.9  # td NaN for q<=0; absurd q
    p[16], p[17] = 0.0, -1.0             # theta = inf / NaN
    q[18], r[18] = 0.085, 99.0           # very hot and humid: Bolton Newton breakdown region
    td[18] = 322.5
    q[19], td[19], r[19] = 0.04, 309.5, 98.0
    p[21], p[22] = 1000.0, 106000.0      # model top / below sea level
    r[12], r[13] = 0.0, 100.0
    return dict(t=np.array(t), td=td, r=r, q=q, p=p)


def run_cases(dataset, base, manifest, store, dtypes=(np.float64, np.float32)):
    for dt in dtypes:
        tag = "f64" if dt is np.float64 else "f32"
        pool = derived_pool(base, dt)
        for k, v in pool.items():
            store[f"{dataset}.{tag}.in.{k}"] = v
        for func, args, kw in case_table():
            cid = f"{dataset}.{tag}.{func}" + "".join(f".{v}" for v in kw.values())
            try:
                out = getattr(ref, func)(*[pool[a].copy() for a in args], **kw)
            except Exception as exc:  # recorded, so the product can be checked for the same behaviour
                manifest.append(dict(id=cid, dataset=dataset, dtype=tag, func=func, args=args, kwargs=kw,
                                     raises=type(exc).__name__))
                continue
            outs = out if isinstance(out, tuple) else (out,)
            for i, o in enumerate(outs):
                store[f"{cid}.out{i}"] = np.asarray(o)
            manifest.append(dict(id=cid, dataset=dataset, dtype=tag, func=func, args=args, kwargs=kw,
                                 nout=len(outs), out_dtype=[str(np.asarray(o).dtype) for o in outs]))


def main():
    manifest, store = [], {}

    # (1) the reference's own 480-row input table
    d = read_csv("t_hum_p_data.csv")
    run_cases("csv480", {k: d[k] for k in ("t", "td", "r", "q", "p")}, manifest, store)

    # (2) seeded physical sample across the whole column (benchmark distribution)
    t, q, p, _ = synthetic.make_fields(16, 48, dtype=np.float64, seed=synthetic.SEED)
    t, q, p = t.ravel(), q.ravel(), p.ravel()
    rng = np.random.default_rng(7)
    r = ref.relative_humidity_from_specific_humidity(t, q, p)
    td = ref.dewpoint_from_specific_humidity(q, p)
    td = np.minimum(td, t - rng.uniform(0.0, 0.5, size=t.shape))
    run_cases("synth768", dict(t=t, td=td, r=r, q=q, p=p), manifest, store)

    # (3) edge cases
    run_cases("edge", edge_base(), manifest, store)

    # (4) broadcasting / N-D / scalar / list behaviour of the direct functions
    t2 = np.linspace(220.0, 300.0, 5 * 7).reshape(5, 7)
    q2 = np.linspace(1e-5, 0.01, 5 * 7).reshape(5, 7)
    pl = np.array([20000.0, 50000.0, 70000.0, 85000.0, 100000.0])
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        a, b, c = t2.astype(dt), q2.astype(dt), pl.astype(dt)
        store[f"bcast.{tag}.in.t"], store[f"bcast.{tag}.in.q"], store[f"bcast.{tag}.in.pl"] = a, b, c
        store[f"bcast.{tag}.theta_levmajor"] = ref.potential_temperature(a, c[:, None])
        store[f"bcast.{tag}.rh_levmajor"] = ref.relative_humidity_from_specific_humidity(a, b, c[:, None])
        store[f"bcast.{tag}.theta_levminor"] = ref.potential_temperature(a.T.copy(), c)
        store[f"bcast.{tag}.td_scalar_p"] = ref.dewpoint_from_specific_humidity(b, dt(85000.0))
        store[f"bcast.{tag}.ept_levmajor"] = ref.ept_from_specific_humidity(a, b, c[:, None])
        store[f"bcast.{tag}.wb_newton_2d"] = ref.wet_bulb_temperature_from_specific_humidity(
            a, b, np.broadcast_to(c[:, None], a.shape).copy(), t_method="newton")
    store["scalar.theta"] = np.asarray(ref.potential_temperature(264.12, 85000.0))
    store["scalar.es"] = np.asarray(ref.saturation_vapour_pressure(np.float64(300.0)))
    store["list.theta"] = np.asarray(ref.potential_temperature([264.12, 261.45], [85000, 85000]))
    store["readme.theta"] = ref.potential_temperature(np.array([264.12, 261.45]), np.array([85000.0, 85000.0]))

    # call signatures of the 39 public functions (names, order, defaults) as data
    import inspect
    sigs = {n: str(inspect.signature(getattr(ref, n))) for n in dir(ref)
            if not n.startswith("_") and inspect.isfunction(getattr(ref, n)) and n != "array_namespace"}
    store["signatures"] = np.frombuffer(json.dumps(sigs).encode(), dtype=np.uint8)
    store["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "thermo_golden.npz"), **store)

    # (5) the reference's CSV fixtures as data
    csv = {}
    for name in ("eqpt", "sat_mr", "sat_mr_slope", "sat_q", "sat_q_slope", "sat_vp", "sat_vp_slope", "seqpt",
                 "t_hum_p_data", "t_on_most_adiabat", "t_wet", "t_wetpt"):
        d = read_csv(name + ".csv")
        for col in d.dtype.names:
            csv[f"{name}.{col}"] = np.asarray(d[col])
    np.savez_compressed(os.path.join(HERE, "ref_csv.npz"), **csv)

    nraise = sum(1 for m in manifest if "raises" in m)
    print(f"{len(manifest)} cases ({nraise} raising), {len(store)} arrays;",
          os.path.getsize(os.path.join(HERE, 'thermo_golden.npz')) // 1024, "KiB +",
          os.path.getsize(os.path.join(HERE, 'ref_csv.npz')) // 1024, "KiB")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for `vertical.pressure_on_hybrid_levels`, recorded from the REFERENCE
(build container only; stand-ins for the un-vendored earthkit-utils and deprecation packages in
tests/golden/_standin; the reference's own 153 hybrid-level tests pass on them).

Writes tests/golden/vertical_golden.npz: the IFS L137/L91 A/B tables the reference ships
(conf/ifs_levels_conf.json through hybrid_level_parameters), the data of its fixture
tests/vertical/_hybrid_core_data.py, and recorded outputs for several sp fields, dtypes, level
selections and alpha_top variants.  Data only.
"""
import importlib.util
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("EKM_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "_standin"), os.path.join(REF, "src")]

from earthkit.meteo.vertical import array as ref  # noqa: E402

warnings.simplefilter("ignore")
np.seterr(all="ignore")


def main():
    store, manifest = {}, []
    for n in (137, 91):
        A, B = ref.hybrid_level_parameters(n)
        store[f"coef.{n}.A"], store[f"coef.{n}.B"] = A, B

    spec = importlib.util.spec_from_file_location("core", os.path.join(REF, "tests", "vertical", "_hybrid_core_data.py"))
    core = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(core)
    for k in ("A", "B", "p_surf", "p_full", "p_half", "delta", "alpha"):
        store[f"fixture.{k}"] = np.asarray(getattr(core, k))

    rng = np.random.default_rng(20260313)
    sps = {"pts2": np.asarray(core.p_surf), "rand1d": rng.uniform(50000.0, 105000.0, 131),
           "rand2d": rng.uniform(50000.0, 105000.0, (3, 5)), "scalar": np.asarray(98765.4)}
    cases = []
    for nlev in (137, 91):
        for sp_name in sps:
            for dt in ("f64", "f32"):
                cases.append(dict(nlev=nlev, sp=sp_name, dtype=dt, levels=None, alpha_top="ifs",
                                  output=["full", "half", "delta", "alpha"], vertical_axis=0))
    for levels in ([137], [1], [1, 2, 3], [135, 136, 137], [137, 136, 135], [1, 137], [40, 96, 41], [5, 5, 7]):
        for out in (["full"], ["half"], ["delta", "alpha"], ["half", "full"]):
            cases.append(dict(nlev=137, sp="rand1d", dtype="f64", levels=levels, alpha_top="ifs", output=out,
                              vertical_axis=0))
    cases.append(dict(nlev=137, sp="rand1d", dtype="f64", levels=None, alpha_top="arpege",
                      output=["delta", "alpha"], vertical_axis=0))
    cases.append(dict(nlev=137, sp="rand2d", dtype="f64", levels=None, alpha_top="ifs", output=["full", "alpha"],
                      vertical_axis=2))
    cases.append(dict(nlev=137, sp="rand2d", dtype="f32", levels=[100, 120], alpha_top="ifs", output=["full"],
                      vertical_axis=1))
    # a table whose top half-level pressure is > 0.1 Pa (the non-TOA branch, vertical.py:683-684, 698-701)
    cases.append(dict(nlev=137, sp="rand1d", dtype="f64", levels=None, alpha_top="ifs", top_offset=5.0,
                      output=["full", "half", "delta", "alpha"], vertical_axis=0))
    for k, v in sps.items():
        store[f"sp.{k}"] = v
    for i, c in enumerate(cases):
        A, B = store[f"coef.{c['nlev']}.A"].copy(), store[f"coef.{c['nlev']}.B"].copy()
        if c.get("top_offset"):
            A = A + c["top_offset"]
        npdt = np.float32 if c["dtype"] == "f32" else np.float64
        sp = sps[c["sp"]].astype(npdt)
        if c["dtype"] == "f32":
            A, B = A.astype(npdt), B.astype(npdt)
        res = ref.pressure_on_hybrid_levels(A, B, sp, levels=c["levels"], alpha_top=c["alpha_top"],
                                            output=c["output"], vertical_axis=c["vertical_axis"])
        res = res if isinstance(res, tuple) else (res,)
        c["id"] = f"case{i:03d}"
        c["out_dtype"] = [str(r.dtype) for r in res]
        for name, r in zip(c["output"], res):
            store[f"{c['id']}.{name}"] = np.asarray(r)
        manifest.append(c)
    store["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "vertical_golden.npz")
    np.savez_compressed(path, **store)
    print(len(manifest), "cases,", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

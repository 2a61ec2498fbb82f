#!/usr/bin/env python3
"""Refresh profiles/valu_latest.json (what bench.py falls back to when its own in-run counter pass is unavailable) from
bench.py lines whose VALU side was measured in the run (`roofline.valu_counters`, e.g. tools/bench_matrix.sh ... --valu
measure, tools/bench_f64.sh):

    python tools/update_valu_latest.py gpurun_out/r04/matrix.jsonl gpurun_out/r04/bench_f64.jsonl
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = {"pipeline_full": "full", "pipeline_svp_td_rh": "p3", "potential_temperature": "theta", "saturation_vapour_pressure": "svp",
      "relative_humidity_from_specific_humidity": "rh", "ept_from_specific_humidity": "ept"}


def main():
    path = os.path.join(ROOT, "profiles", "valu_latest.json")
    cur = json.load(open(path)) if os.path.exists(path) else {"counts_per_point": {}}
    n = 0
    for f in sys.argv[1:]:
        for ln in open(f):
            if not ln.startswith("{"):
                continue
            d = json.loads(ln)
            r = d.get("roofline") or {}
            if not str(r.get("valu_source", "")).startswith("measured in this run"):
                continue
            entry = d["config"]["entry_point"][4:-4]
            wl = WL.get(entry)
            if wl is None and entry == "wet_bulb_temperature_from_specific_humidity":
                desc = d["config"]["workload"]
                wl = "wetbulb" if "newton" in desc else ("wetbulb_bisect" if "(ifs" in desc else "wetbulb_bisect_" + desc.split("(")[1].split(",")[0])
            if wl is None:
                continue
            cur["counts_per_point"][f"{wl}:{d['config']['p_mode']}:{d['dtype']}"] = r["valu_counters"]
            n += 1
    cur["source"] = ("executed VALU wave-instructions x 64 lanes / points by class, measured by bench.py's own rocprofv3 --pmc child "
                     "pass (roofline.valu_counters of tools/bench_matrix.sh --valu measure and tools/bench_f64.sh lines)")
    json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    print(f"{n} entries refreshed, {len(cur['counts_per_point'])} in {path}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense the SQ counter pass of tools/profile_valu.sh into one JSON entry.

    python tools/summarize_valu.py gpurun_out/valu_wetbulb wetbulb profiles/r02_valu_counters.json

Per point: VALU wave-instructions x 64 lanes / points; transcendentals issue at a quarter of the plain rate,
so issue units = VALU + 3 x transcendental."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, key, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.loads([ln for ln in open(os.path.join(src, "bench_valu.json")) if ln.startswith("{")][-1])
    npts = bench["roofline"]["points_per_launch"]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, "pmc_valu", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the benchmarked kernel: the map_* kernel with the most VALU instructions in total
    name = max((k for k in agg if "map_" in k or "columns" in k), key=lambda k: sum(agg[k].get("SQ_INSTS_VALU", [0])))
    raw = {c: sum(v) / len(v) for c, v in agg[name].items()}
    valu, trans = raw["SQ_INSTS_VALU"] * 64 / npts, raw.get("SQ_INSTS_VALU_TRANS_F32", 0.0) * 64 / npts
    out = {"kernel": name[:160], "points_per_launch": npts, "kernel_ms_in_this_pass": bench["roofline"]["kernel_ms"],
           "valu_instr_per_point": valu, "trans_instr_per_point": trans, "issue_units_per_point_f32": valu + 3 * trans,
           "valu_active_over_wave_cycles": raw.get("SQ_ACTIVE_INST_VALU", 0) / raw["SQ_WAVE_CYCLES"],
           "wait_any_over_wave_cycles": raw.get("SQ_WAIT_ANY", 0) / raw["SQ_WAVE_CYCLES"],
           "wait_inst_any_over_wave_cycles": raw.get("SQ_WAIT_INST_ANY", 0) / raw["SQ_WAVE_CYCLES"], "raw": raw}
    cur = json.load(open(dst)) if os.path.exists(dst) else {}
    cur[key] = out
    json.dump(cur, open(dst, "w"), indent=1)
    # what bench.py falls back to when its own in-run counter pass is unavailable (profiles/valu_latest.json): the executed
    # wave-instructions per point by class; bench.py::valu_units weighs them
    args = bench["config"]
    bkey = f"{key.split('@')[0]}:{args['p_mode']}:{bench['dtype']}"
    lat = os.path.join(os.path.dirname(dst), "valu_latest.json")
    v = json.load(open(lat)) if os.path.exists(lat) else {}
    v.setdefault("counts_per_point", {})
    v["source"] = ("executed VALU wave-instructions x 64 lanes / points by class from rocprofv3 --pmc passes of bench.py "
                   "(tools/profile_valu.sh)")
    v["counts_per_point"][bkey] = {c: round(x * 64 / npts, 3) for c, x in raw.items() if c.startswith("SQ_INSTS_VALU")}
    json.dump(v, open(lat, "w"), indent=1)
    print(key, json.dumps({k: v for k, v in out.items() if k != "raw"}))


if __name__ == "__main__":
    main()

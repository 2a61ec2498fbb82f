#!/bin/bash
# fp64 bench lines (one field of 3600x1800x137 fp64 points; P5, P3, wet-bulb Newton / bisection, theta, rh, ept),
# optionally with every lane forced through the plain-double pass (EKM_F64_PLAIN=1) as the A/B of the two-pass scheme.
#   tools/bench_f64.sh <out.jsonl> [extra bench.py args...]
OUT=${1:?out.jsonl}; shift
: > "$OUT"
for wl in full p3 wetbulb wetbulb_bisect wetbulb_bisect_bolton35 wetbulb_bisect_bolton39 theta rh ept svp; do
  timeout -k 10 150 python3 bench.py --workload $wl --dtype f64 --steps 10 --warmup 3 --no-cpu-baseline --traffic none --valu measure --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err" || echo "{\"failed\": \"$wl\"}" >> "$OUT"
done
for pm in level hybrid; do
  timeout -k 10 150 python3 bench.py --workload full --pmode $pm --dtype f64 --steps 10 --warmup 3 --no-cpu-baseline --traffic none --valu measure --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err" || echo "{\"failed\": \"full $pm\"}" >> "$OUT"
done
python3 - "$OUT" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "failed" in d:
        print("FAILED", d["failed"]); continue
    r, c = d["roofline"], d["config"]
    print(f"{c['entry_point'][4:48]:46s} {c['p_mode']:6s} {d['dtype']} {r['kernel_ms']:8.3f} ms  hbm frac {r.get('hbm_frac', r['frac']):.3f}  "
          f"parity {d['parity']['ok'] if d['parity'] else None} maxrel {d['parity']['max_rel_err'] if d['parity'] else None}")
PY

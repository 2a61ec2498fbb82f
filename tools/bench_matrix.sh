#!/bin/bash
# One bench.py line per (workload, pressure mode): the roofline table of DESIGN.md section 4.
#   tools/bench_matrix.sh <out.jsonl> [extra bench.py args...]
OUT=${1:?out.jsonl}; shift
: > "$OUT"
sleep_between=${EKM_MATRIX_SLEEP:-6}  # seconds between runs: a bench process returns 40-100 GB of device memory at exit, and while the driver clears it every host<->device copy on the GPU runs at half rate (profiles/r06_host_path_rate.txt)
for wl in full p3 wetbulb wetbulb_bisect theta rh ept; do
  for pm in field level hybrid; do
    timeout -k 10 120 python3 bench.py --workload $wl --pmode $pm --steps 20 --warmup 5 --no-cpu-baseline --traffic file --valu file --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err" || echo "{\"failed\": \"$wl $pm\"}" >> "$OUT"; sleep $sleep_between
  done
done
for wl in wetbulb_bisect_bolton35 wetbulb_bisect_bolton39; do
  timeout -k 10 120 python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --traffic file --valu file --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err" || echo "{\"failed\": \"$wl\"}" >> "$OUT"
done
timeout -k 10 120 python3 bench.py --workload svp --steps 20 --warmup 5 --no-cpu-baseline --traffic file --valu file --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err"
timeout -k 10 120 python3 bench.py --workload hybrid_levels --steps 20 --warmup 5 --traffic file --valu file --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err"
timeout -k 10 120 python3 bench.py --workload geopotential --steps 20 --warmup 5 --traffic file --valu file --sustain 0 "$@" >> "$OUT" 2>> "$OUT.err"
python3 - "$OUT" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "failed" in d:
        print("FAILED", d["failed"]); continue
    r, c = d["roofline"], d["config"]
    hbm = r.get("hbm_achieved_gbs", r["achieved"])
    sets = ""
    if r.get("kernel_ms_sets"):  # the same kernel on three independently allocated buffer sets (bench.py --buffer-sets)
        f = [r["bytes_per_point"] * r["points_per_launch"] / (m * 1e-3) / 1e9 / 8000.0 for m in r["kernel_ms_sets"]]
        sets = f"  sets hbm frac min/med/max {min(f):.3f}/{sorted(f)[len(f) // 2]:.3f}/{max(f):.3f}"
    e2e = d.get("end_to_end")
    sets += f"  e2e {e2e['call_ms']:.1f} ms {e2e['gbs']:.0f} GB/s" if e2e else ""
    pts = d["parity"].get("points") if d["parity"] else None
    print(f"{c['workload'].split(' on ')[0][:44]:44s} {c['p_mode']:6s} {d['dtype']} {r['bytes_per_point']:3d} B/pt  {r['kernel_ms']:7.3f} ms  "
          f"{hbm:7.1f} GB/s  hbm frac {r.get('hbm_frac', r['frac']):.3f}  " + (f"valu frac {r['frac']:.3f}  " if r["bound"] == "valu" else "") + f"parity {d['parity']['ok'] if d['parity'] else None} "
          f"maxrel {d['parity']['max_rel_err'] if d['parity'] else None} on {pts} points" + sets)
PY

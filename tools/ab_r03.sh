#!/bin/bash
# In-process A/B of the current library against the round-3 library (built from commit 2b653d0 into
# earthkit-meteo_amd/variants/r03/): same buffers, interleaved rounds (tools/sweep.py), default launch tuning in both.
#   tools/ab_r03.sh <outdir>
O=${1:?outdir}; mkdir -p "$O"
for pm in field level hybrid; do
  timeout -k 10 300 python3 tools/sweep.py --libs default,variants/r03/libekm_thermo.so --workloads full,p3,wetbulb,wetbulb_bisect,theta --tiles 0 --unroll 0 \
    --rounds 5 --steps 5 --pmode $pm --out "$O/ab_f32_$pm.json" > "$O/ab_f32_$pm.txt" 2>&1 || echo "failed $pm"
done
timeout -k 10 300 python3 tools/sweep.py --dtype f64 --libs default,variants/r03/libekm_thermo.so --workloads full,p3,wetbulb,wetbulb_bisect,theta --tiles 0 --unroll 0 \
  --rounds 3 --steps 3 --out "$O/ab_f64_field.json" > "$O/ab_f64_field.txt" 2>&1 || echo "failed f64"
for f in "$O"/ab_*.txt; do echo "== $f"; grep -v "^$" "$f" | cut -c1-28,60-200; done

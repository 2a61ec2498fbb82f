#!/bin/bash
# Per-depth attribution of the tree walk's LDS cycles: the fp32 IFS bisection wet-bulb built with the
# walk stopped after d = 2, 4, ... 12 steps (earthkit-meteo_amd/variants/d<d>: -DEKM_WALK_DEPTH=d; results are garbage, the
# counters are what is wanted), one rocprofv3 --pmc pass each (the program right after `--`, the variant picked through the
# environment), field-mode pressure.  Differences between successive depths = what those two steps cost.
#   tools/pmc_bisect_depth.sh <outdir>
O=${1:?outdir}; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
fail=0
for d in 2 4 6 8 10 12; do
  if [ $d -eq 12 ]; then unset EKM_THERMO_LIB; else export EKM_THERMO_LIB=$GRAFT_REPO_ROOT/earthkit-meteo_amd/variants/d$d/libekm_thermo.so; fi
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc_d${d}_$i" -- python3 bench.py --workload wetbulb_bisect --pmode field --steps 3 --warmup 1 --no-cpu-baseline --traffic none --valu none --sustain 0 --no-stream-ceiling --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 > "$O/out_d${d}_$i.txt" 2> "$O/err_d${d}_$i.txt"
    rc=$?
    if [ $rc -ne 0 ]; then echo "pmc_bisect_depth: pass d=$d/$i FAILED rc=$rc"; tail -5 "$O/err_d${d}_$i.txt"; fail=1; fi
    i=$((i+1))
  done
done
unset EKM_THERMO_LIB
python3 - "$O" <<'PY'
import csv, glob, collections, sys
n = 887760000
rows = {}
for d in (2, 4, 6, 8, 10, 12):
    agg = collections.defaultdict(list)
    dur = []
    for dd in sorted(glob.glob(f"{sys.argv[1]}/pmc_d{d}_*")):
        for f in glob.glob(dd + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "map_" in r["Kernel_Name"] and "OpWetBulb" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows[d] = {c: (sum(x) / len(x) * 64 / n if c.startswith("SQ_INSTS") else sum(x) / len(x) / 1e6) for c, x in agg.items()}
keys = ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
print("walk stopped after d steps (SQ_INSTS_* per point, the rest in millions of cycles per launch)")
print("  d " + " ".join(f"{k[3:]:>18s}" for k in keys))
prev = None
for d in (2, 4, 6, 8, 10, 12):
    print(f"{d:3d} " + " ".join(f"{rows[d].get(k, float('nan')):18.2f}" for k in keys))
print("per two steps (difference to the row above):")
for a, b in ((2, 4), (4, 6), (6, 8), (8, 10), (10, 12)):
    print(f"{a:2d}-{b:<2d}" + " ".join(f"{rows[b].get(k, float('nan')) - rows[a].get(k, float('nan')):18.2f}" for k in keys))
PY
exit $fail

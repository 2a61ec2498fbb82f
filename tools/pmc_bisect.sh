#!/bin/bash
# SQ / LDS counters of the bisection wet-bulb kernel (one rocprofv3 pass per counter set, the program right after `--`).
#   tools/pmc_bisect.sh <outdir> [extra bench.py flags]
# A failing pass is reported with its stderr, never skipped silently (the old script compared against a
# round-3 library through EKM_THERMO_LIB, which the new binding could not load, and lost that column without a word;
# libraries of another round are refused by their ABI version now -- A/B against one goes through tools/sweep.py).
O=${1:?outdir}; shift; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SETS=("SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES")
fail=0
for pm in field level; do
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc_${pm}_$i" -- python3 bench.py --workload wetbulb_bisect --pmode $pm --steps 3 --warmup 1 --no-cpu-baseline --traffic none --valu none --sustain 0 --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 "$@" > "$O/out_${pm}_$i.txt" 2> "$O/err_${pm}_$i.txt"
    rc=$?
    if [ $rc -ne 0 ]; then echo "pmc_bisect: pass $pm/$i ($set) FAILED rc=$rc"; tail -5 "$O/err_${pm}_$i.txt"; fail=1; fi
    i=$((i+1))
  done
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
for pm in ("field", "level"):
    agg = collections.defaultdict(list)
    for d in sorted(glob.glob(f"{sys.argv[1]}/pmc_{pm}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "map_" in r["Kernel_Name"] and "OpWetBulb" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n = 887760000
    print(pm, {c: round(sum(x) / len(x) * 64 / n, 2) if c.startswith("SQ_INSTS") else round(sum(x) / len(x) / 1e6, 1) for c, x in sorted(agg.items())})
PY
exit $fail

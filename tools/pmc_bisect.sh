#!/bin/bash
# SQ / LDS counters of the bisection wet-bulb kernel, current library and (if built) the round-3 library.
#   tools/pmc_bisect.sh <outdir>
O=${1:?outdir}; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in new r03; do
  lib=""; [ $tag = r03 ] && lib=earthkit-meteo_amd/variants/r03/libekm_thermo.so
  [ $tag = r03 ] && [ ! -f $lib ] && continue
  for pm in field level; do
    EKM_THERMO_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      --kernel-trace --output-format csv -d "$O/pmc_${tag}_${pm}" -- python3 bench.py --workload wetbulb_bisect --pmode $pm --steps 3 --warmup 1 --no-cpu-baseline --traffic none --valu none --sustain 0 > /dev/null 2> "$O/err_${tag}_${pm}.txt"
  done
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob(sys.argv[1] + "/pmc_*")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "map_" in k and "OpWetBulb" in k:
            n = 887760000
            print(d.split("/")[-1], {c: round(sum(x) / len(x) * 64 / n, 2) if c.startswith("SQ_INSTS") else round(sum(x) / len(x) / 1e6, 1) for c, x in sorted(v.items())})
PY

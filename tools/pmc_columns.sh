#!/bin/bash
# Memory-side counters of the column kernels (geopotential_columns, hybrid_levels) beside a map kernel of the same
# 12 B/point class (theta: 8 B read + 4 B written) and P3: one rocprofv3 --pmc pass per counter set, each a run of its own.
#   tools/pmc_columns.sh <outdir>
O=${1:?outdir}; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  for wl in geopotential hybrid_levels theta p3; do
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pass${i}_$wl" -- python3 bench.py --workload $wl --steps 3 --warmup 1 \
      --no-cpu-baseline --traffic none --valu none --sustain 0 --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 > "$O/pass${i}_$wl.json" 2> "$O/pass${i}_$wl.err" || echo "pass $i $wl failed"
  done
done
python3 - "$O" <<'PY'
import collections, csv, glob, json, os, sys
O = sys.argv[1]
table = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(O, "pass*_*"))):
    if not os.path.isdir(d):
        continue
    wl = os.path.basename(d).split("_", 1)[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ks = [k for k in agg if any(t in k for t in ("map_", "geopotential_columns", "hybrid_levels", "hybrid_rows")) and "fill" not in k]
    if not ks:
        continue
    k = max(ks, key=lambda k: len(next(iter(agg[k].values()))))
    for c, v in agg[k].items():
        table[c][wl] = sum(v) / len(v)
    try:
        ms = json.loads([ln for ln in open(d + ".json") if ln.startswith("{")][0])["roofline"]["kernel_ms"]
        table["kernel_ms (in that pass)"].setdefault(wl, ms)
    except Exception:
        pass
wls = ["geopotential", "hybrid_levels", "theta", "p3"]
print(f"{'counter (mean per launch)':44s}" + "".join(f"{w:>16s}" for w in wls))
for c in sorted(table):
    print(f"{c:44s}" + "".join(f"{table[c].get(w, float('nan')):16.5g}" for w in wls))
PY

#!/bin/bash
# VALU-side counters of one bench.py workload (own rocprofv3 pass: --pmc with --kernel-trace only).
#   tools/profile_valu.sh <outdir> [bench.py args...]
set -u
O=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$O"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES ${EKM_EXTRA_PMC:-} --kernel-trace --output-format csv -d "$O/pmc_valu" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --traffic none --valu none --sustain 0 --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 "$@" > "$O/bench_valu.json" 2> "$O/valu.err" || exit 1
echo "valu counters in $O"

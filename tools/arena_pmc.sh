#!/bin/bash
# tools/arena_pmc.sh <outdir>: the counter passes of tools/arena_pmc.py (each --pmc set in a run of its own)
O=${1:?outdir}; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/arena_pmc.py > "$O/timing.txt" 2>&1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pass$i" -- python3 tools/arena_pmc.py > "$O/pass$i.txt" 2>&1 || echo "pass $i failed"
  echo "== pass $i: $set"; cat "$O/pass$i.txt" | grep "round 1"; python3 tools/arena_pmc.py --summarize "$O/pass$i"
done

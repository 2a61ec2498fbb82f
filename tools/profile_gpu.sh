#!/bin/bash
# Run on the GPU box (through gpurun): three rocprofv3 passes of one bench.py workload.
#   tools/profile_gpu.sh <outdir> [bench.py args...]
# kernel-trace + stats in one pass; FETCH_SIZE and WRITE_SIZE each in a pass of their own
# (they do not fit one pass on gfx950).  The program follows `--` directly (no wrapper).
set -u
O=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$O"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --traffic file --valu file --sustain 0 --no-stream-ceiling --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 "$@" > "$O/bench_trace.json" 2> "$O/trace.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --traffic file --valu file --sustain 0 --no-stream-ceiling --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 "$@" > "$O/bench_fetch.json" 2> "$O/fetch.err" || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --traffic file --valu file --sustain 0 --no-stream-ceiling --parity-slab-levels 0 --no-end-to-end --buffer-sets 1 "$@" > "$O/bench_write.json" 2> "$O/write.err" || exit 3
echo "profiled into $O"

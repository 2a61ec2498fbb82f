#!/bin/bash
# The rocprofv3 evidence of one round, produced on the GPU box (through gpurun) and summarised into profiles/:
#   tools/profile_round.sh r02 [regex]
# kernel-trace + stats, FETCH_SIZE, WRITE_SIZE (separate passes) for the bench default (P5), P3, wet-bulb, bisect,
# the level / hybrid pressure modes; SQ VALU counters for the VALU-bound kernels.
set -u
R=${1:?round tag}
G=gpurun_out/prof_$R
mkdir -p "$G" profiles
ONLY=${2:-.}   # optional second argument: a regular expression selecting the workloads of this call (a call is limited to 20 minutes)
run() {  # name  summarize-key  bench args...
  local name=$1 key=$2; shift 2
  [[ $name =~ $ONLY ]] || return 0
  tools/profile_gpu.sh "$G/$name" "$@" > "$G/$name.log" 2>&1 || { echo "profile $name failed ($?)"; return 1; }
  python3 tools/summarize_profile.py "$G/$name" "profiles/${R}_$name" "$key" > /dev/null && echo "profiled $name"
}
run full            full:field:f32:137
run p3              p3:field:f32:137            --workload p3
run wetbulb         wetbulb:field:f32:137       --workload wetbulb
run bisect          wetbulb_bisect:field:f32:137 --workload wetbulb_bisect
run bisect_f64      wetbulb_bisect:field:f64:137 --workload wetbulb_bisect --dtype f64
run bisect_bolton35 wetbulb_bisect_bolton35:field:f32:137 --workload wetbulb_bisect_bolton35
run full_level      full:level:f32:137          --pmode level
run full_hybrid     full:hybrid:f32:137         --pmode hybrid
run p3_level        p3:level:f32:137            --workload p3 --pmode level
run p3_hybrid       p3:hybrid:f32:137           --workload p3 --pmode hybrid
run wetbulb_hybrid  wetbulb:hybrid:f32:137      --workload wetbulb --pmode hybrid
run theta_hybrid    theta:hybrid:f32:137        --workload theta --pmode hybrid
run geopotential    geopotential:hybrid:f32:137 --workload geopotential
run hybrid_levels   hybrid_levels:hybrid:f32:137 --workload hybrid_levels
run bisect_bolton39 wetbulb_bisect_bolton39:field:f32:137 --workload wetbulb_bisect_bolton39
run bisect_bolton35_f64 wetbulb_bisect_bolton35:field:f64:137 --workload wetbulb_bisect_bolton35 --dtype f64
run full_f64        full:field:f64:137          --dtype f64
run wetbulb_f64     wetbulb:field:f64:137       --workload wetbulb --dtype f64
finish() { mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* profiles/traffic_latest.json profiles/valu_latest.json gpurun_out/profiles_$R/; }
[[ valu =~ $ONLY ]] || { finish; exit 0; }
for wl in full wetbulb wetbulb_bisect wetbulb_bisect_bolton35 wetbulb_bisect_bolton39 p3; do
  tools/profile_valu.sh "$G/valu_$wl" --workload $wl > "$G/valu_$wl.log" 2>&1 && python3 tools/summarize_valu.py "$G/valu_$wl" $wl "profiles/${R}_valu_counters.json"
done
for wl in wetbulb wetbulb_bisect; do
  tools/profile_valu.sh "$G/valu_${wl}_level" --workload $wl --pmode level > /dev/null 2>&1 && python3 tools/summarize_valu.py "$G/valu_${wl}_level" ${wl}@level "profiles/${R}_valu_counters.json"
done
finish

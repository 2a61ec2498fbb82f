#!/bin/bash
# pressure_on_hybrid_levels (p_full of the 3600x1800x137 field): the row-ordered kernel against the column kernel,
# alternating in fresh processes.  Usage: tools/ab_hybrid_rows.sh [out.txt]
out=${1:-gpurun_out/ab_hybrid_rows.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for dtype in f32 f64; do
  for r in 1 0 1 0; do
    EKM_HYBRID_ROWS=$r python bench.py --workload hybrid_levels --dtype $dtype --steps 20 --warmup 3 --sustain 0 --traffic none --valu none --no-cpu-baseline 2>/dev/null \
      | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$dtype hybrid_rows=$r  %.4f ms  frac %.3f  parity %s' % (d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['ms_per_step'], d['roofline']['frac'], d.get('parity', {}).get('ok')))" >> "$out" || exit 1
  done
done
cat "$out"

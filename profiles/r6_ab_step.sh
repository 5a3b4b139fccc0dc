#!/bin/bash
# same-box A/B of the headline step on an environment switch: profiles/r6_ab_step.sh VAR A B [reps]
V=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq 1 $R); do
  for x in $A $B; do
    echo "$V=$x: $(env $V=$x python bench.py --no-cpu-baseline --no-f32-exact --no-matcher-active --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s")')"
  done
done

#!/bin/bash
# same-box comparison of several builds of the library on the headline step: profiles/r6_ab_libs.sh reps lib1.so lib2.so ...
R=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    echo "$L: $(TDR_LIB_PATH=$PWD/textualdegremoval_amd/$L python bench.py --no-cpu-baseline --no-f32-exact --no-matcher-active --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), "ms")')"
  done
done

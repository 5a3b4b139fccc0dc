#!/bin/bash
# round 5: marginal wall cost of kernel families inside the captured bx3 step (profiles/probe_ablate.py), same box
tag=$1; shift
mkdir -p gpurun_out/r5
for abl in "" "$@" ""; do
  ABL=$abl timeout 300 python profiles/probe_ablate.py 2>&1 | tail -1
done | tee gpurun_out/r5/ablate_$tag.log

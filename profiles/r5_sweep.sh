#!/bin/bash
# round 5: same-box A/B of the default (bx3) step under tuning knobs; one line per setting.  usage: r5_sweep.sh <tag> "<ENV=.. ENV=..>" ...
tag=$1; shift
mkdir -p gpurun_out/r5
run() { t=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', round(d['ms_per_step'],2))"; }
( for cfg in "$@"; do run "$cfg" $cfg; done ) | tee gpurun_out/r5/sweep_$tag.log

#!/bin/bash
# round 5: same-box A/B of the default (bx3) step under tuning knobs; one line per setting
mkdir -p gpurun_out/r5
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2))"; }
( run base A=1
  run p16_min32 TDR_P16_MIN_C=32
  run wg_want256 TDR_WG_WANT=256
  run wg_want768 TDR_WG_WANT=768
  run wgp_want256 TDR_WGP_WANT=256
  run wgp_want1024 TDR_WGP_WANT=1024
  run nodefer TDR_DEFER_WGRAD=0
  run noln_defer TDR_DEFER_LN_FINISH=0
  run base2 A=1 ) | tee gpurun_out/r5/sweep_$1.log

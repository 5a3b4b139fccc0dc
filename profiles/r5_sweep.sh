#!/bin/bash
# round 5: same-box A/B of the default (bx3) step under tuning knobs; one line per setting
mkdir -p gpurun_out/r5
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2))"; }
( run base A=1
  run old_defaults TDR_P16_MIN_C=64 TDR_WG_WANT=512
  run dwf_rpt8 TDR_DWF_RPT=8
  run dwf_rpt4 TDR_DWF_RPT=4
  run wg_want128 TDR_WG_WANT=128
  run wgp_want384 TDR_WGP_WANT=384
  run s2_want128 TDR_WG_S2_WANT=128
  run s2_want512 TDR_WG_S2_WANT=512
  run base2 A=1 ) | tee gpurun_out/r5/sweep_$1.log

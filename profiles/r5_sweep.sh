#!/bin/bash
# round 5: same-box A/B of the default (bx3) step under tuning knobs; one line per setting
mkdir -p gpurun_out/r5
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2))"; }
( run staged TDR_WG1=0
  run dma_want512 TDR_WG1=1 TDR_WG1_WANT=512
  run dma_want256 TDR_WG1=1 TDR_WG1_WANT=256
  run dma_want384 TDR_WG1=1 TDR_WG1_WANT=384
  run staged2 TDR_WG1=0 ) | tee gpurun_out/r5/sweep_$1.log

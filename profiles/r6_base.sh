#!/bin/bash
# round-6 baseline on the box at hand: default line (short), DP schedule, kernel summary of the replayed step
mkdir -p gpurun_out/r6
python bench.py --no-cpu-baseline --no-f32-exact --no-matcher-active --no-roofline > gpurun_out/r6/base_default.log 2>&1
TDR_FORCE_DP_SCHEDULE=1 python bench.py --no-cpu-baseline --no-f32-exact --no-matcher-active --no-roofline > gpurun_out/r6/base_dp.log 2>&1
bash profiles/rocprof_run.sh gpurun_out/r6/base_kernel_summary.txt 27 -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active
tail -1 gpurun_out/r6/base_default.log | cut -c1-300; tail -1 gpurun_out/r6/base_dp.log | cut -c1-300; head -3 gpurun_out/r6/base_kernel_summary.txt

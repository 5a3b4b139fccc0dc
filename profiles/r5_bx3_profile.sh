#!/bin/bash
# round 5: kernel summary of the reference-arithmetic (TDR_MATH=bx3) step, as found at the start of the round
mkdir -p gpurun_out/r5
export TDR_MATH=bx3
bash profiles/rocprof_run.sh gpurun_out/r5/rocprofv3_bx3_steps_$1.txt 27 -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active
cp /tmp/tdr_prof_cmd.log gpurun_out/r5/bench_bx3_under_rocprof_$1.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active > gpurun_out/r5/bench_bx3_$1.log 2>&1
tail -1 gpurun_out/r5/bench_bx3_$1.log | cut -c1-300
head -40 gpurun_out/r5/rocprofv3_bx3_steps_$1.txt

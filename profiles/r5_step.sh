#!/bin/bash
# round 5 iteration helper: <tag> [pytest args...] -- fused-NAF tests, then the bx3 bench line and its kernel summary
tag=$1; shift
mkdir -p gpurun_out/r5
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu > gpurun_out/r5/pytest_$tag.log 2>&1; tail -5 gpurun_out/r5/pytest_$tag.log; fi
export TDR_MATH=bx3
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active > gpurun_out/r5/bench_bx3_$tag.log 2>&1
tail -1 gpurun_out/r5/bench_bx3_$tag.log | cut -c1-200
bash profiles/rocprof_run.sh gpurun_out/r5/rocprofv3_bx3_steps_$tag.txt 27 -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active
head -45 gpurun_out/r5/rocprofv3_bx3_steps_$tag.txt

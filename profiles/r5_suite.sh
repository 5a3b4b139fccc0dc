#!/bin/bash
# round 5: full -m gpu suite + the default bench line (all legs) on the current tree
tag=$1
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r5/pytest_gpu_$tag.log 2>&1
tail -25 gpurun_out/r5/pytest_gpu_$tag.log
python bench.py > gpurun_out/r5/bench_default_$tag.log 2>&1
tail -1 gpurun_out/r5/bench_default_$tag.log | cut -c1-400

#!/bin/bash
# One gpurun call that regenerates the round-2 evidence under gpurun_out/r2m/ (copied to profiles/r2/ afterwards):
#   PMC traffic (two --pmc passes), rocprofv3 kernel summaries (bench command with and without its roofline leg),
#   the default bench line, the secondary bench lines, smoke() and the full -m gpu suite.
mkdir -p gpurun_out/r2m profiles/r2
bash profiles/pmc_collect.sh gpurun_out/r2m/pmc_traffic.json > gpurun_out/r2m/pmc_collect.log 2>&1
cp gpurun_out/r2m/pmc_traffic.json profiles/r2/pmc_traffic.json
bash profiles/rocprof_run.sh gpurun_out/r2m/rocprofv3_kernel_summary_bench.txt 18 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact
cp /tmp/tdr_prof_cmd.log gpurun_out/r2m/bench_under_rocprof.log
bash profiles/rocprof_run.sh gpurun_out/r2m/rocprofv3_kernel_summary_steps.txt 27 -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline
python bench.py > gpurun_out/r2m/bench_default.log 2>&1
python bench.py --arch restormer --no-cpu-baseline --no-f32-exact > gpurun_out/r2m/bench_restormer_cfg3.log 2>&1
python bench.py --arch restormer --size 512 --batch 2 --no-cpu-baseline --no-f32-exact > gpurun_out/r2m/bench_restormer_cfg5.log 2>&1
TDR_MATH=h1 python bench.py --arch restormer --size 512 --batch 2 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r2m/bench_restormer_cfg5_h1.log 2>&1
python bench.py --arch promptir --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r2m/bench_promptir_384_bs8.log 2>&1
python bench.py --arch drsformer --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r2m/bench_drsformer_256_bs8.log 2>&1
python bench.py --arch drsformer_mefc --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r2m/bench_drsformer_mefc_256_bs8.log 2>&1
python bench.py --width 64 --size 384 --batch 8 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r2m/bench_nafnet_yaml_w64_384_bs8.log 2>&1
python bench.py --dino-ref-size 640 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r2m/bench_dino640.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2m/smoke.log 2>&1
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r2m/pytest_gpu.log
for f in bench_default bench_restormer_cfg3 bench_restormer_cfg5 bench_restormer_cfg5_h1 bench_promptir_384_bs8 bench_drsformer_256_bs8 bench_drsformer_mefc_256_bs8 bench_nafnet_yaml_w64_384_bs8 bench_dino640; do echo "$f: $(tail -1 gpurun_out/r2m/$f.log | cut -c1-170)"; done
tail -1 gpurun_out/r2m/smoke.log; tail -2 gpurun_out/r2m/pytest_gpu.log; head -2 gpurun_out/r2m/rocprofv3_kernel_summary_steps.txt

#!/bin/bash
# One gpurun call that regenerates the round-4 evidence under gpurun_out/r4m/ (copied to profiles/r4/ afterwards):
#   PMC traffic (two --pmc passes, tagged with the hash of the conv kernel source), rocprofv3 kernel summaries (bench command with and
#   without its roofline leg; the stage-A step), the default bench line, the secondary bench lines, smoke() and the full -m gpu suite.
mkdir -p gpurun_out/r4m profiles/r4
rm -rf gpurun_out/margins
bash profiles/pmc_collect.sh gpurun_out/r4m/pmc_traffic.json > gpurun_out/r4m/pmc_collect.log 2>&1
cp gpurun_out/r4m/pmc_traffic.json profiles/r4/pmc_traffic.json
bash profiles/rocprof_run.sh gpurun_out/r4m/rocprofv3_kernel_summary_bench.txt 18 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact --no-matcher-active
cp /tmp/tdr_prof_cmd.log gpurun_out/r4m/bench_under_rocprof.log
bash profiles/rocprof_run.sh gpurun_out/r4m/rocprofv3_kernel_summary_steps.txt 27 -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active
bash profiles/rocprof_run.sh gpurun_out/r4m/rocprofv3_i2t_step_summary.txt 16 -- python /root/repo/bench.py --arch i2t --steps 10 --warmup 2
bash profiles/rocprof_run.sh gpurun_out/r4m/rocprofv3_dino640_summary.txt 18 -- python /root/repo/bench.py --dino-ref-size 640 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline
python bench.py > gpurun_out/r4m/bench_default.log 2>&1
python bench.py --arch i2t --steps 10 --warmup 2 > gpurun_out/r4m/bench_i2t_step.log 2>&1
python bench.py --arch i2t --clip L --steps 10 --warmup 2 > gpurun_out/r4m/bench_i2t_step_vit_l14.log 2>&1
python bench.py --arch tr --steps 10 --warmup 2 > gpurun_out/r4m/bench_tr_step.log 2>&1
TDR_P16=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active > gpurun_out/r4m/bench_p16_off.log 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active > gpurun_out/r4m/bench_p16_on.log 2>&1
python bench.py --arch restormer --no-cpu-baseline --no-f32-exact > gpurun_out/r4m/bench_restormer_cfg3.log 2>&1
python bench.py --arch restormer --size 512 --batch 2 --no-cpu-baseline --no-f32-exact > gpurun_out/r4m/bench_restormer_cfg5.log 2>&1
python bench.py --arch promptir --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r4m/bench_promptir_384_bs8.log 2>&1
python bench.py --arch drsformer --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r4m/bench_drsformer_256_bs8.log 2>&1
python bench.py --arch drsformer_mefc --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r4m/bench_drsformer_mefc_256_bs8.log 2>&1
python bench.py --dino-ref-size 640 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r4m/bench_dino640.log 2>&1
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r4m/bench_2ranks_gloo_one_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4m/smoke.log 2>&1
cp -r gpurun_out/margins gpurun_out/r4m/margins 2>/dev/null
for f in bench_default bench_p16_on bench_p16_off bench_tr_step bench_i2t_step bench_i2t_step_vit_l14 bench_restormer_cfg3 bench_restormer_cfg5 bench_promptir_384_bs8 bench_drsformer_256_bs8 bench_drsformer_mefc_256_bs8 bench_dino640 bench_2ranks_gloo_one_gpu; do echo "$f: $(tail -1 gpurun_out/r4m/$f.log | cut -c1-170)"; done
tail -1 gpurun_out/r4m/smoke.log; head -2 gpurun_out/r4m/rocprofv3_kernel_summary_steps.txt

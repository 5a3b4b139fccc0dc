#!/bin/bash
# One gpurun call that regenerates the round-6 evidence under gpurun_out/r6m/ (copied to profiles/r6/ afterwards), all in the DEFAULT
# arithmetic (TDR_MATH=bx3: 3-way bf16 split, 24-bit operands, fp32 range, no loss scale / guard):
#   PMC traffic (two --pmc passes, tagged with the hashes of the conv kernel sources), SQ counters of every kernel of an eager step,
#   rocprofv3 kernel summaries (bench command with and without its roofline leg), the timeline of the replayed step, the default bench
#   line (all legs), the secondary bench lines, smoke().
mkdir -p gpurun_out/r6m profiles/r6
rm -rf gpurun_out/margins
bash profiles/pmc_collect.sh gpurun_out/r6m/pmc_traffic.json > gpurun_out/r6m/pmc_collect.log 2>&1
cp gpurun_out/r6m/pmc_traffic.json profiles/r6/pmc_traffic.json
bash profiles/pmc_run.sh gpurun_out/r6m/sq_a.txt "kernel" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" -- python /root/repo/profiles/pmc_workload.py
bash profiles/pmc_run.sh gpurun_out/r6m/sq_b.txt "kernel" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" -- python /root/repo/profiles/pmc_workload.py
bash profiles/pmc_clock.sh gpurun_out/r6m/pmc_clock.txt -- python /root/repo/profiles/pmc_workload.py
cp gpurun_out/r6m/pmc_clock.txt.json profiles/r6/pmc_clock.json
bash profiles/rocprof_run.sh gpurun_out/r6m/rocprofv3_kernel_summary_bench.txt 18 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact --no-matcher-active
cp /tmp/tdr_prof_cmd.log gpurun_out/r6m/bench_under_rocprof.log
bash profiles/rocprof_run.sh gpurun_out/r6m/rocprofv3_kernel_summary_steps.txt 27 -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active
db=$(find /tmp/tdr_prof -name '*.db' | head -1)
n=$(head -1 gpurun_out/r6m/rocprofv3_kernel_summary_steps.txt | sed 's/.*; \([0-9]*\) launches.*/\1/')
python profiles/timeline_gaps.py "$db" "$n" 5 > gpurun_out/r6m/timeline_gaps.txt 2>&1
python bench.py > gpurun_out/r6m/bench_default.log 2>&1
python bench.py --arch restormer --no-cpu-baseline --no-f32-exact > gpurun_out/r6m/bench_restormer_cfg3.log 2>&1
python bench.py --arch restormer --size 512 --batch 2 --no-cpu-baseline --no-f32-exact > gpurun_out/r6m/bench_restormer_cfg5_h1.log 2>&1
python bench.py --arch restormer --size 512 --batch 2 --math bx3 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r6m/bench_restormer_cfg5_bx3.log 2>&1
bash profiles/rocprof_run.sh gpurun_out/r6m/rocprofv3_restormer_cfg5_h1_steps.txt 17 -- python /root/repo/bench.py --arch restormer --size 512 --batch 2 --steps 10 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline
TDR_FORCE_DP_SCHEDULE=1 python bench.py --no-cpu-baseline --no-f32-exact --no-matcher-active --no-roofline > gpurun_out/r6m/bench_dp_schedule_one_gpu.log 2>&1
TDR_FORCE_COLLECTIVES=1 python bench.py --rccl-dry-run > gpurun_out/r6m/rccl_dry_run_one_rank.log 2>&1
python bench.py --arch promptir --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r6m/bench_promptir_384_bs8.log 2>&1
python bench.py --arch drsformer --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r6m/bench_drsformer_256_bs8.log 2>&1
python bench.py --arch drsformer_mefc --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r6m/bench_drsformer_mefc_256_bs8.log 2>&1
python bench.py --dino-ref-size 640 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r6m/bench_dino640.log 2>&1
python bench.py --arch i2t --steps 10 --warmup 2 > gpurun_out/r6m/bench_i2t_step.log 2>&1
python bench.py --arch tr --steps 10 --warmup 2 > gpurun_out/r6m/bench_tr_step.log 2>&1
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline > gpurun_out/r6m/bench_2ranks_gloo_one_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6m/smoke.log 2>&1
for f in bench_default bench_restormer_cfg3 bench_restormer_cfg5_h1 bench_restormer_cfg5_bx3 bench_dp_schedule_one_gpu rccl_dry_run_one_rank bench_promptir_384_bs8 bench_drsformer_256_bs8 bench_drsformer_mefc_256_bs8 bench_dino640 bench_i2t_step bench_tr_step bench_2ranks_gloo_one_gpu; do echo "$f: $(tail -1 gpurun_out/r6m/$f.log | cut -c1-170)"; done
tail -1 gpurun_out/r6m/smoke.log; head -2 gpurun_out/r6m/rocprofv3_kernel_summary_steps.txt; head -3 gpurun_out/r6m/timeline_gaps.txt

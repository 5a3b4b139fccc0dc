mkdir -p gpurun_out/r2m
bash profiles/pmc_collect.sh gpurun_out/r2m/pmc_traffic.json > gpurun_out/r2m/pmc_collect.log 2>&1
cp gpurun_out/r2m/pmc_traffic.json profiles/r2/pmc_traffic.json
bash profiles/rocprof_run.sh gpurun_out/r2m/rocprofv3_kernel_summary_r2b.txt 18 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact > gpurun_out/r2m/bench_under_rocprof_r2b.log 2>&1
cp /tmp/tdr_prof_cmd.log gpurun_out/r2m/bench_under_rocprof_r2b.log
python bench.py > gpurun_out/r2m/bench_default_r2b.log 2>&1
python bench.py --arch restormer --no-cpu-baseline --no-f32-exact > gpurun_out/r2m/bench_restormer_cfg3_r2b.log 2>&1
python bench.py --arch restormer --size 512 --batch 2 --no-cpu-baseline --no-f32-exact > gpurun_out/r2m/bench_restormer_cfg5_r2b.log 2>&1
tail -1 gpurun_out/r2m/bench_default_r2b.log | cut -c1-400
tail -1 gpurun_out/r2m/bench_restormer_cfg3_r2b.log | cut -c1-250
tail -1 gpurun_out/r2m/bench_restormer_cfg5_r2b.log | cut -c1-250
head -3 gpurun_out/r2m/rocprofv3_kernel_summary_r2b.txt

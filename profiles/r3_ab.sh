python -m pytest tests/test_hip_kernels.py tests/test_hip_network.py tests/test_hip_step.py -m gpu -x -q 2>&1 | tail -2
for r in 0 1; do echo "headline RING3=$r: $(TDR_RING3=$r python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | cut -c130-175)"; done
for r in 0 1; do echo "restormer RING3=$r: $(TDR_RING3=$r python bench.py --arch restormer --steps 5 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline 2>&1 | tail -1 | cut -c130-175)"; done
for r in 0 1; do echo "promptir RING3=$r: $(TDR_RING3=$r python bench.py --arch promptir --steps 3 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline 2>&1 | tail -1 | cut -c130-175)"; done

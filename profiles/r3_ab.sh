for mb in 0 64 16; do echo "== TDR_CONV_NT_MB=$mb"; TDR_CONV_NT_MB=$mb python profiles/probe_conv1x1_thin.py 2>&1 | tail -10 | head -8; done
for mb in 0 64 0 64; do echo "restormer NT=$mb: $(TDR_CONV_NT_MB=$mb python bench.py --arch restormer --steps 5 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline 2>&1 | tail -1 | cut -c130-175)"; done
for mb in 0 64 0 64; do echo "headline NT=$mb: $(TDR_CONV_NT_MB=$mb python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | cut -c130-175)"; done

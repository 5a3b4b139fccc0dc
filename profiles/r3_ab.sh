python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "wgrad" 2>&1 | tail -1
for i in 1 2; do
echo "base: $(TDR_LIB_PATH=$PWD/profiles/ab/libtdr_hip_base.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | cut -c130-175)"
echo "new : $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | cut -c130-175)"
done

python -m pytest tests/test_hip_kernels.py tests/test_hip_i2t.py -m gpu -x -q -k "conv or clip" 2>&1 | tail -1
for i in 1 2; do
echo "base: $(TDR_LIB_PATH=$PWD/profiles/ab/libtdr_hip_base.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | cut -c130-175)"
echo "new : $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | cut -c130-175)"
done
echo "i2t base: $(TDR_LIB_PATH=$PWD/profiles/ab/libtdr_hip_base.so python bench.py --arch i2t --steps 20 --warmup 3 2>&1 | tail -1 | cut -c150-200)"
echo "i2t new : $(python bench.py --arch i2t --steps 20 --warmup 3 2>&1 | tail -1 | cut -c150-200)"
echo "i2t base: $(TDR_LIB_PATH=$PWD/profiles/ab/libtdr_hip_base.so python bench.py --arch i2t --steps 20 --warmup 3 2>&1 | tail -1 | cut -c150-200)"
echo "i2t new : $(python bench.py --arch i2t --steps 20 --warmup 3 2>&1 | tail -1 | cut -c150-200)"

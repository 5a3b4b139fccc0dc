export TDR_LIB_PATH=$PWD/textualdegremoval_amd/libtdr_hip_tuning.so TDR_FORCE_DP_SCHEDULE=1
run() { python bench.py --no-cpu-baseline --no-f32-exact --no-matcher-active --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), "ms")'; }
for i in 1 2; do
  echo "base: $(run)"
  echo "GRP_BIG=1: $(TDR_WG1_GRP_BIG=1 run)"
  echo "SP=1: $(TDR_WG1_SP=1 run)"
  echo "GRP_BIG=1 SP=1: $(TDR_WG1_GRP_BIG=1 TDR_WG1_SP=1 run)"
done

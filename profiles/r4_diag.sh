mkdir -p gpurun_out/r4d
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tdr_prof
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/tdr_prof -o t -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active > /tmp/cmd.log 2>&1
db=$(find /tmp/tdr_prof -name '*.db' | head -1)
cd $root
python profiles/timeline_gaps.py "$db" 1501 5 > gpurun_out/r4d/timeline_gaps.txt 2>&1
timeout 1500 python profiles/diag_bias_grad.py 512 > gpurun_out/r4d/diag_bias_grad.log 2>&1
tail -5 gpurun_out/r4d/timeline_gaps.txt; tail -8 gpurun_out/r4d/diag_bias_grad.log

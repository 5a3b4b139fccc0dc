python -m pytest tests/test_hip_dino.py -m gpu -x -q -s 2>&1 | tail -15
python bench.py --dino-ref-size 640 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline 2>&1 | tail -1 | cut -c1-200
TDR_DINO_TOK16=0 python bench.py --dino-ref-size 640 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline 2>&1 | tail -1 | cut -c1-200
bash profiles/rocprof_run.sh gpurun_out/r3m/rocprofv3_dino640_summary.txt 18 -- python /root/repo/bench.py --dino-ref-size 640 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline
grep -n "tok_\|transpose_f32\|attn_fwd" gpurun_out/r3m/rocprofv3_dino640_summary.txt | head -20

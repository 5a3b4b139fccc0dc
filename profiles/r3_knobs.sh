# one-box A/B of launch-geometry knobs on the headline step (each line: knob, ms/step)
cd /root/repo
for cfg in "base:" "wg_want_256:TDR_WG_WANT=256" "wg_want_384:TDR_WG_WANT=384" "side_wgrad:TDR_SIDE_WGRAD=1" "base2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-exact --no-roofline --no-matcher-active 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3))"
done

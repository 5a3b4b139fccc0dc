cd /root/repo
for cfg in "default:" "wpf_off:TDR_C1_WPF_STAGES=1000" "cfg3_wpf768:TDR_BX_CFG1=3 TDR_C1_BLOCKS=768 TDR_C1_WPF_BLOCKS=768" "cfg2_wpf:TDR_BX_CFG1=2 TDR_C1_BLOCKS=768 TDR_C1_WPF_BLOCKS=768" "cfg3_nowpf:TDR_BX_CFG1=3 TDR_C1_BLOCKS=768 TDR_C1_WPF_STAGES=1000"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name ($envs)"
  env $envs python profiles/bench_i2t.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,2) if isinstance(v,float) else v for k,v in d.items() if 'vit' in k or 'mapper' in k})"
done

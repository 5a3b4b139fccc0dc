for cfg in 0 1; do for want in 128 256 512 1024; do
echo "cfg $cfg want $want"; TDR_WGB_CFG1X1=$cfg TDR_WG_WANT=$want python profiles/probe_wgrad.py 2>&1 | grep "1x1"
done; done

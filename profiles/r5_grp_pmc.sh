#!/bin/bash
# round 5: L2 hit rate and HBM fetch of the grouped 1x1 weight-gradient launch alone
mkdir -p gpurun_out/r5
python profiles/probe_wgrad1x1_group.py 2>&1 | grep grouped | tee gpurun_out/r5/probe_wgrad1x1_group.log
bash profiles/pmc_run.sh gpurun_out/r5/grp_pmc_a.txt "wgrad1x1_sp" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" -- python /root/repo/profiles/probe_wgrad1x1_group.py
bash profiles/pmc_run.sh gpurun_out/r5/grp_pmc_b.txt "wgrad1x1_sp" "FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" -- python /root/repo/profiles/probe_wgrad1x1_group.py
cat gpurun_out/r5/grp_pmc_a.txt gpurun_out/r5/grp_pmc_b.txt

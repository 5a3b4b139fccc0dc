#!/bin/bash
# round 5: L2 hit rate / HBM fetch of the 1x1 weight-gradient kernels at a long contraction (do the tiles of a split share L2 ?)
mkdir -p gpurun_out/r5
export TDR_MATH=bx3 TDR_WG1_WANT=256
for sp in 1 0; do
  TDR_WG1_SP=$sp bash profiles/pmc_run.sh gpurun_out/r5/longk_pmc_sp${sp}_a.txt "wgrad1x1" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" -- python /root/repo/profiles/probe_wgrad1x1_longk.py
  TDR_WG1_SP=$sp bash profiles/pmc_run.sh gpurun_out/r5/longk_pmc_sp${sp}_b.txt "wgrad1x1" "FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python /root/repo/profiles/probe_wgrad1x1_longk.py
  TDR_WG1_SP=$sp bash profiles/pmc_run.sh gpurun_out/r5/longk_pmc_sp${sp}_c.txt "wgrad1x1" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python /root/repo/profiles/probe_wgrad1x1_longk.py
done
cat gpurun_out/r5/longk_pmc_sp*.txt

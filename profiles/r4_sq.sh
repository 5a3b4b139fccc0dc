#!/bin/bash
# SQ counters (wave residency, wait / MFMA-busy / VALU / LDS shares) of the 3x3 kernels of one eager step, round 4: two --pmc passes
mkdir -p gpurun_out/r4d
bash profiles/pmc_run.sh gpurun_out/r4d/sq_a.txt "3" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" -- python /root/repo/profiles/pmc_workload.py
bash profiles/pmc_run.sh gpurun_out/r4d/sq_b.txt "3" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python /root/repo/profiles/pmc_workload.py
grep -A 9 "conv3x3_p16_kernel\|wgrad3x3_p16_kernel\|conv_bx3_kernel<3, 1, 1, 1, 2" gpurun_out/r4d/sq_a.txt | head -80

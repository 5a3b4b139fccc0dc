#!/bin/bash
# round 5: SQ counters (wave residency, wait / MFMA-busy / VALU / LDS shares) of EVERY kernel of one eager step in the default arithmetic (bx3)
mkdir -p gpurun_out/r5
bash profiles/pmc_run.sh gpurun_out/r5/sq_a.txt "kernel" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" -- python /root/repo/profiles/pmc_workload.py
bash profiles/pmc_run.sh gpurun_out/r5/sq_b.txt "kernel" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" -- python /root/repo/profiles/pmc_workload.py
wc -l gpurun_out/r5/sq_a.txt gpurun_out/r5/sq_b.txt

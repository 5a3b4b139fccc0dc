#!/bin/bash
# SQ / cache counters of tok_gemm_kernel<NPL 3> (TDR_TOK3_STAGE = 2 and 3) -> gpurun_out/r5/pmc_tok16x3.txt
mkdir -p gpurun_out/r5
out=gpurun_out/r5/pmc_tok16x3.txt; : > $out
for st in 2 3; do
  export TDR_TOK3_STAGE=$st
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
    echo "== TDR_TOK3_STAGE=$st : $set" >> $out
    bash profiles/pmc_run.sh /tmp/pmc_one.txt "tok_gemm_kernel" "$set" -- python /root/repo/profiles/pmc_tok16x3_workload.py
    cat /tmp/pmc_one.txt >> $out
    tail -3 /tmp/tdr_pmc_cmd.log | grep -i "error\|invalid" >> $out
  done
done
cat $out

#!/bin/bash
# round 5: split-once 1x1 weight gradient (wgrad1x1_sp_kernel) against the LDS-DMA kernel, error vs fp64 + time per launch
mkdir -p gpurun_out/r5
( for sp in 1 0; do TDR_MATH=bx3 TDR_WG1_SP=$sp python profiles/probe_wgrad1x1.py 2>&1 | grep -v amdgpu.ids | sed "s/^/SP=$sp /"; done
  TDR_MATH=hx2 TDR_WG1_SP=1 python profiles/probe_wgrad1x1.py 2>&1 | grep -v amdgpu.ids | sed "s/^/SP=1 /" ) | tee gpurun_out/r5/probe_wgrad1x1_sp_$1.log

#!/bin/bash
# round 5: steady-state cost per (workgroup, 32-pixel stage) of the 1x1 weight-gradient kernels at a long contraction
mkdir -p gpurun_out/r5
( for m in bx3 hx2; do for sp in 1 0; do TDR_MATH=$m TDR_WG1_SP=$sp TDR_WG1_WANT=256 python profiles/probe_wgrad1x1_longk.py 2>&1 | grep want | sed "s/^/$m SP=$sp /"; done; done ) | tee gpurun_out/r5/probe_wgrad1x1_longk_$1.log

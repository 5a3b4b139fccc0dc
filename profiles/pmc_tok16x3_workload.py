#!/usr/bin/env python
"""counter workload: tdr_tok16x3_gemm at the matcher's fc1 shape (27 520 x 3072 x 768, GELU -> planes), 6 launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('bx3')
torch.manual_seed(0)
P, N, Kd = 20 * 1376, 3072, 768
x, w, bias = torch.randn(P, Kd, device='cuda'), torch.randn(N, Kd, device='cuda') * 0.05, torch.randn(N, device='cuda')
x3, w3 = K.split_planes3(x), K.split_planes3(w)
for _ in range(6):
    K.tok16x3_gemm(x3, w3, bias, epi=4, act=2)
torch.cuda.synchronize()

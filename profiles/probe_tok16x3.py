#!/usr/bin/env python
"""Token-major GEMM on bf16 triple planes (tdr_tok16x3_gemm, the default 'bx3' arithmetic) at the DINOv2 matcher's shapes (ref 640^2: 20 images x
1376 padded token rows, ViT-B/14), against the channel-major engine (conv_bx3_kernel) on the same problem.  TF = fp32-equivalent (2 P N K / t);
the 6-product ceiling is 417.  usage: [TDR_TOK3_STAGE=0|1|2] python profiles/probe_tok16x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('bx3')
torch.manual_seed(0)
P = 20 * 1376


def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, Kd, epi in (('qkv', 2304, 768, 3), ('proj', 768, 768, 2), ('fc1', 3072, 768, 4), ('fc2', 768, 3072, 2)):
    x, w, bias = torch.randn(P, Kd, device='cuda'), torch.randn(N, Kd, device='cuda') * 0.05, torch.randn(N, device='cuda')
    x3, w3 = K.split_planes3(x), K.split_planes3(w)
    res = torch.zeros(P, N, device='cuda')
    f = lambda: K.tok16x3_gemm(x3, w3, bias, epi=epi, act=2 if epi == 4 else 0, out32=res if epi == 2 else None)
    us = timeit(f)
    xc = x.t().contiguous().view(1, Kd, P // 32, 32)
    wp, mp, *_ = K.pack_weights(w.view(N, Kd, 1, 1).contiguous(), K.PACK_FWD)
    g = lambda: K.conv_forward(xc, wp, mp, N, 1, bias=bias, relu=2 if epi == 4 else 0)
    usc = timeit(g)
    flop = 2.0 * P * N * Kd
    print(f'{name:5s} P {P} N {N} K {Kd}: planes {us:7.1f} us ({flop / us * 1e-6:4.0f} TF)   channel-major {usc:7.1f} us ({flop / usc * 1e-6:4.0f} TF)', flush=True)
t = torch.randn(P, 768, device='cuda')
w, b = torch.randn(768, device='cuda'), torch.randn(768, device='cuda')
print(f'LayerNorm -> 3 planes {timeit(lambda: K.tok_layernorm(t, w, b, 1e-6, planes=3)):.1f} us; fp32 {timeit(lambda: K.tok_layernorm(t, w, b, 1e-6, out_f16=False)):.1f} us')
a = torch.randn(768, P, device='cuda')
print(f'channel-major -> 3 planes {timeit(lambda: K.cm_to_tok16x3(a)):.1f} us')
qkv = torch.randn(1, 2304, P // 32, 32, device='cuda')
print(f'attention (tdr_attention_fwd_math, bx3) {timeit(lambda: K.attention_fwd(qkv, 12, 0.125, 1370, flat_batch=20)):.1f} us')

#!/usr/bin/env python
"""Numerics of Winograd F(2x2, 3x3) for the MASA-encoder 3x3 layers, CPU only (the next lever on the 16.6 ms of plane convolutions).
Same data through (a) a direct fp32 convolution and (b) F(2x2, 3x3) with fp32 transforms and fp32 element-wise GEMMs, both against float64; the
element-wise products of (b) are also run on 3-way bf16-split operands (6 products, fp32 accumulate) as the library's kernels would.
Prints max / rms errors relative to the output's rms.  usage: python profiles/probe_winograd_numerics.py"""
import torch

torch.manual_seed(0)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split3(x):
    h = x.to(torch.bfloat16).float()
    r = x - h
    m = r.to(torch.bfloat16).float()
    return h, m, (r - m).to(torch.bfloat16).float()


def mm_split(a, b):
    """sum_k a[..k] b[k..] with 3-way bf16 split operands, six products, fp32 accumulation (small terms first)"""
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = al @ bh + ah @ bl
    acc = acc + am @ bm
    acc = acc + am @ bh + ah @ bm
    return acc + ah @ bh


def winograd(x, w, dt, mm):
    N, C, H, W = x.shape
    Ko = w.shape[0]
    U = torch.einsum('ij,kcjl,ml->imkc', G.to(dt), w.to(dt), G.to(dt))                      # [4,4,K,C]
    xp = torch.nn.functional.pad(x.to(dt), (1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                                    # [N,C,th,tw,4,4]
    V = torch.einsum('ij,nctwjl,ml->imnctw', Bt.to(dt), t, Bt.to(dt))                         # [4,4,N,C,th,tw]
    th, tw = V.shape[4], V.shape[5]
    Vm = V.permute(0, 1, 3, 2, 4, 5).reshape(4, 4, C, N * th * tw)
    M = torch.stack([torch.stack([mm(U[i, j], Vm[i, j]) for j in range(4)]) for i in range(4)])   # [4,4,K,P]
    Y = torch.einsum('ij,jlkp,ml->imkp', At.to(dt), M, At.to(dt))                             # [2,2,K,P]
    Y = Y.reshape(2, 2, Ko, N, th, tw).permute(3, 2, 4, 0, 5, 1).reshape(N, Ko, 2 * th, 2 * tw)
    return Y


for C, H in ((64, 64), (128, 32), (256, 32)):
    x = torch.relu(torch.randn(2, C, H, H)) * 0.7                       # post-ReLU activations, like the ResidualBlock inputs
    w = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    rms = ref.pow(2).mean().sqrt().item()
    direct = torch.nn.functional.conv2d(x, w, padding=1)
    xs, ws = x.reshape(1, -1), None
    # direct convolution on split operands: im2col GEMM
    cols = torch.nn.functional.unfold(x, 3, padding=1)                  # [N, C*9, HW]
    dsplit = torch.stack([mm_split(w.reshape(C, -1), cols[n]) for n in range(2)]).reshape(2, C, H, H)
    wf32 = winograd(x, w, torch.float32, lambda a, b: a @ b)
    wsplit = winograd(x, w, torch.float32, mm_split)
    w64 = winograd(x, w, torch.float64, lambda a, b: a @ b)
    e = lambda y: ((y.double() - ref).abs().max().item() / rms, (y.double() - ref).pow(2).mean().sqrt().item() / rms)
    print(f'C {C} @{H}x{H}: rel-to-rms max / rms error  direct fp32 {e(direct)[0]:.2e} / {e(direct)[1]:.2e}   direct 3xbf16 {e(dsplit)[0]:.2e} / {e(dsplit)[1]:.2e}   '
          f'Winograd fp32 {e(wf32)[0]:.2e} / {e(wf32)[1]:.2e}   Winograd 3xbf16 {e(wsplit)[0]:.2e} / {e(wsplit)[1]:.2e}   (Winograd fp64 {e(w64)[0]:.1e})', flush=True)

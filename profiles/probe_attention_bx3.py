"""The frozen ViTs' attention at the default arithmetic: tdr_attention_fwd_math code 1 (3-way bf16 split, attn_fwd_bx3_kernel) against the exact
fp32 MFMA kernel (code 0) and a float64 reference, at the DINOv2 matcher's shape (16 windows x 4 images -> B = 64 passes of 12 heads x 1370 tokens,
head dim 64) and CLIP ViT-H's (head dim 80, 257 tokens).  python profiles/probe_attention_bx3.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from textualdegremoval_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()


def run(qkv, heads, scale, T, math):
    B, C3, LD = qkv.shape
    out = torch.empty(B, C3 // 3, LD, device='cuda')
    _lib.check(lib.tdr_attention_fwd_math(qkv.data_ptr(), B, C3 // 3, heads, T, LD, float(scale), math, 0, out.data_ptr(), K._stream()), 'attn')
    return out


for (B, heads, hd, T) in ((16, 12, 64, 1370), (8, 16, 80, 257), (4, 2, 32, 300)):
    LD = (T + 31) // 32 * 32
    g = torch.Generator().manual_seed(B + hd)
    qkv = (torch.randn(B, 3 * heads * hd, LD, generator=g) * 1.5).cuda()
    scale = hd ** -0.5
    outs = {m: run(qkv, heads, scale, T, m) for m in (0, 1)}
    q, k, v = (t.double().cpu().view(B, heads, hd, LD)[..., :T] for t in qkv.chunk(3, 1))
    ref = torch.einsum('bhqk,bhdk->bhdq', torch.softmax(torch.einsum('bhdq,bhdk->bhqk', q, k) * scale, -1), v).reshape(B, heads * hd, T)
    errs = {m: (outs[m].cpu().double()[..., :T] - ref).abs().max().item() for m in outs}
    ts = {}
    for m in (0, 1):
        for _ in range(3):
            run(qkv, heads, scale, T, m)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(10):
            run(qkv, heads, scale, T, m)
        e[1].record()
        torch.cuda.synchronize()
        ts[m] = e[0].elapsed_time(e[1]) / 10 * 1e3
    fl = 4.0 * B * heads * T * T * hd
    print(f'B {B} heads {heads} hd {hd} T {T}: exact fp32 {ts[0]:.0f} us ({fl / ts[0] / 1e6:.0f} TF), bx3 {ts[1]:.0f} us ({fl / ts[1] / 1e6:.0f} TF); '
          f'max |err| vs float64: exact {errs[0]:.2e}, bx3 {errs[1]:.2e}', flush=True)

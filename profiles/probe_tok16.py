"""Times the token-major fp16 kernels of the DINOv2 matcher (csrc/tdr_tok16.hip) at the matcher-active shapes of the headline
workload (24 images x 1312 padded tokens, ViT-B/14): us per launch and TFLOP/s.  python profiles/probe_tok16.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

B, LD, T1, D = 24, 1312, 1297, 768
P = B * LD
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
for name, N, Kd, epi in (('qkv', 2304, 768, 0), ('proj', 768, 768, 2), ('fc1', 3072, 768, 1), ('fc2', 768, 3072, 2)):
    x, w, bias, ls, res = r(P, Kd).half(), (r(N, Kd) * 0.03).half(), r(N), r(N), r(P, N)
    us = timeit(lambda: K.tok16_gemm(x, w, bias, epi=epi, res=res if epi == 2 else None, ls=ls if epi == 2 else None))
    print(f'{name:5s} P={P} N={N} K={Kd} epi={epi}: {us:7.1f} us  {2 * P * N * Kd / us / 1e6:6.0f} TFLOP/s')
qkv = r(P, 3 * D).half()
us = timeit(lambda: K.tok16_attention(qkv, B, 12, 0.125, T1))
print(f'attention B={B} T={T1}: {us:7.1f} us  {4 * B * 12 * T1 * T1 * 64 / us / 1e6:6.0f} TFLOP/s')
x = r(P, D); w = r(D); b = r(D)
print(f'layernorm -> fp16: {timeit(lambda: K.tok_layernorm(x, w, b, 1e-6)):7.1f} us')

# the 2-way split GEMMs of the stage-A CLIP ViT-H/14 encoder (4 images x 288 padded tokens), weights cold (32 distinct layers cycled)
P, D = 4 * 288, 1280
L = 8
for name, N, Kd, epi in (('qkv', 3 * D, D, 3), ('out', D, D, 2), ('fc1', 4 * D, D, 4), ('fc2', D, 4 * D, 2)):
    x2 = K.split_planes(r(P, Kd))
    ws = [K.split_planes(r(N, Kd) * 0.03) for _ in range(L)]
    bias, res = r(N), r(P, N)
    i = [0]
    def f():
        i[0] = (i[0] + 1) % L
        K.tok16x2_gemm(x2, ws[i[0]], bias, epi=epi, act=2 if epi == 4 else 0, out32=res if epi == 2 else None)
    us = timeit(f, n=40)
    print(f'x2 {name:4s} P={P} N={N} K={Kd}: {us:7.1f} us  {2 * P * N * Kd / us / 1e6:6.0f} TFLOP/s (fp32-equivalent)')

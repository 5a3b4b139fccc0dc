import sys, os
sys.path.insert(0, '/root/repo')
import torch
from textualdegremoval_amd import kernels as K
def timeit(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
D = 1280
for P in (1152, 2304):
  for L in (1, 8):
    for name, N, Kd, epi in (('qkv', 3 * D, D, 3), ('out', D, D, 2), ('fc1', 4 * D, D, 4), ('fc2', D, 4 * D, 2)):
        x2 = K.split_planes(r(P, Kd))
        ws = [K.split_planes(r(N, Kd) * 0.03) for _ in range(L)]
        bias, res = r(N), r(P, N)
        i = [0]
        def f():
            i[0] = (i[0] + 1) % L
            K.tok16x2_gemm(x2, ws[i[0]], bias, epi=epi, act=2 if epi == 4 else 0, out32=res if epi == 2 else None)
        us = timeit(f)
        print(f'P={P} L={L} x2 {name:4s} N={N} K={Kd}: {us:7.1f} us  {2 * P * N * Kd / us / 1e6:6.0f} TF-eq')

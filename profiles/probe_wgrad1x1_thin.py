"""1x1 weight gradients of the C = 48 / 96 Restormer-ref levels (256 x 256, bs 8), fp16 split (loss-scaled backward):
time per launch (reduce included) and algorithmic TB/s.   [TDR_WG_WANT=n] python profiles/probe_wgrad1x1_thin.py"""
import sys
sys.path.insert(0, '.')
import torch
from textualdegremoval_amd import kernels as K


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


N, H = 8, 256
K.set_grad_scaled(True)
for cin, cout, h in [(48, 144, H), (48, 48, H), (48, 254, H), (127, 48, H), (96, 288, H), (96, 96, H), (96, 510, H), (255, 96, H),
                     (192, 1020, H // 2), (510, 192, H // 2), (384, 384, H // 4)]:
    x = torch.randn(N, cin, h, h, device='cuda')
    dy = torch.randn(N, cout, h, h, device='cuda')
    gb = 4 * N * h * h * (cin + cout) / 1e9
    t = bench(lambda: K.conv_wgrad(x, dy, cout, cin, 1))
    print(f'{cin:4d}->{cout:4d} @{h}: wgrad {t:7.1f} us ({gb / t * 1e3:5.2f} TB/s)')

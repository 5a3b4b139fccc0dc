"""1x1 convolutions of the C = 48 / 96 Restormer-ref levels (256 x 256, bs 8): forward, data gradient, weight gradient --
time per launch and algorithmic TB/s (4 * (in + out) bytes).   python profiles/probe_conv1x1_thin.py"""
import sys
sys.path.insert(0, '.')
import torch
from textualdegremoval_amd import engine as E, kernels as K


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
K.set_grad_scaled(True)          # the loss-scaled backward of the default arithmetic: gradients on the fp16 split too
for cin, cout in [(48, 144), (48, 48), (48, 254), (127, 48), (96, 288), (96, 96), (96, 510), (255, 96), (192, 1020), (510, 192)]:
    h = H if cin < 150 or cout < 150 or (cin, cout) in ((96, 288), (96, 510), (255, 96)) else H // 2
    if (cin, cout) in ((192, 1020), (510, 192)):
        h = H // 2
    x = torch.randn(N, cin, h, h, device='cuda')
    w = torch.randn(cout, cin, 1, 1, device='cuda') * 0.1
    dy = torch.randn(N, cout, h, h, device='cuda')
    gb = 4 * N * h * h * (cin + cout) / 1e9
    tf = bench(lambda: E.conv_fwd(x, w, None, 1, 0))
    K.BACKWARD_PHASE = True
    tb = bench(lambda: E.conv_bwd(dy, x, w, 1, 0, bias=False))
    K.BACKWARD_PHASE = False
    print(f'{cin:4d}->{cout:4d} @{h}: fwd {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s)   dgrad+wgrad {tb:7.1f} us ({2 * gb / tb * 1e3:5.2f} TB/s)')

"""DINOv2 ViT-B/14 linears as 1x1 convs over channel-major tokens [B][C][LD/32][32] (16 windows + 4 images of 1370 tokens): time
and fp32-equivalent TFLOP/s per tile configuration.   python profiles/probe_vit_linear.py"""
import os
import sys
sys.path.insert(0, '.')
import torch
from textualdegremoval_amd import _lib, engine as E, kernels as K

CFG1 = {0: 'heuristic', 1: '128x128', 2: '64x256', 3: '64x128', 5: '256x64'}
lib = _lib.load()


def bench(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (16, 4):
    for cin, cout in [(768, 2304), (768, 768), (768, 3072), (3072, 768)]:
        x = torch.randn(B, cin, 43, 32, device='cuda')
        w = torch.randn(cout, cin, 1, 1, device='cuda') * 0.03
        fl = 2 * B * 1376 * cin * cout / 1e12
        row = []
        for c, name in CFG1.items():
            lib.tdr_conv_force_cfg(1, c)
            t = bench(lambda: E.conv_fwd(x, w, None, 1, 0))
            row.append(f'{name} {t:6.1f} us ({fl / t * 1e6:5.0f} TF)')
        lib.tdr_conv_force_cfg(1, 0)
        print(f'B{B} {cin}->{cout}: ' + '  '.join(row))

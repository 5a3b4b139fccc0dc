"""Per-kernel microbenchmarks of the MFMA kernels at cfg2 shapes (HIP-event timing on the launch
stream).  usage: python profiles/microbench.py [filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K

torch.manual_seed(0)
K.WGRAD_1X1_BX3 = True
FILT = sys.argv[1] if len(sys.argv) > 1 else ''
ITERS = int(os.environ.get('ITERS', '10'))
MATHS = os.environ.get('MATHS', 'f32,bx3').split(',')


def timeit(name, fn, flops=None, nbytes=None):
    if FILT and FILT not in name:
        return
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / ITERS
    extra = ''
    if flops:
        extra += f' {flops / ms / 1e9:8.1f} TFLOP/s ({100 * flops / ms / 1e9 / 157.3:4.1f}% f32 MFMA peak)'
    if nbytes:
        extra += f' {nbytes / ms / 1e9:7.2f} TB/s alg'
    print(f'{name:44s} {ms * 1e3:9.1f} us{extra}', flush=True)


def conv_case(tag, N, Cin, Cout, H, KH, stride=1, **kw):
    x = torch.randn(N, Cin, H, H, device='cuda')
    w = torch.randn(Cout, Cin, KH, KH, device='cuda') * 0.05
    b = torch.randn(Cout, device='cuda')
    pad = 1 if KH == 3 else 0
    OH = (H + 2 * pad - KH) // stride + 1
    out = torch.empty(N, Cout, OH, OH, device='cuda')
    fl = 2.0 * N * Cout * Cin * KH * KH * OH * OH
    by = 4.0 * (x.numel() + out.numel())
    outs = {}
    for math in MATHS:
        K.set_math(math)
        wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
        timeit(f'[{math}] conv{KH}x{KH} {tag} N{N} {Cin}->{Cout} @{H}', lambda: K.conv_forward(x, wp, mp, Cout, KH, stride=stride, pad=pad, bias=b, out=out), fl, by)
        outs[math] = out.clone()
    if len(outs) == 2:
        print(f'      max|bx3 - f32| = {(outs["bx3"] - outs["f32"]).abs().max().item():.3e}   (|out| max {outs["f32"].abs().max().item():.2f})')
    go = torch.randn_like(out)
    for math in MATHS:
        K.set_math(math)
        timeit(f'[{math}] wgrad{KH}x{KH} {tag} N{N} {Cin}->{Cout} @{H}', lambda: K.conv_wgrad(x, go, Cout, Cin, KH, stride=stride, pad=pad, want_db=True), fl, by)


if __name__ == '__main__':
    # masa_enc 3x3 (stacked lq+ref batch of 8)
    for lvl, (c, h) in enumerate([(32, 512), (64, 256), (128, 128), (256, 64), (512, 32)]):
        conv_case(f'L{lvl + 1}', 8, c, c, h, 3)
    # NAFBlock 1x1: level 3 backbone (28 blocks), level 0, fusion levels
    conv_case('blk3.conv1', 4, 256, 512, 64, 1)
    conv_case('blk3.conv3', 4, 256, 256, 64, 1)
    conv_case('blk0.conv1', 4, 32, 64, 512, 1)
    conv_case('fus0.conv1', 4, 64, 128, 512, 1)
    conv_case('fus1.conv1', 4, 128, 256, 256, 1)
    conv_case('fus2.conv1', 4, 256, 512, 128, 1)
    conv_case('fus3.conv1', 4, 512, 1024, 64, 1)
    conv_case('fus4.conv1', 4, 1024, 2048, 32, 1)

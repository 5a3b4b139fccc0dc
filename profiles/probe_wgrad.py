"""Ablation timing of wgrad_bx3_kernel (1x1, 2-way fp16 split) -- profiling builds libtdr_wprobeN.so (TDR_WPROBE 1-4,
csrc/tdr_wgrad_bx3.hip).  usage: TDR_LIB_PATH=textualdegremoval_amd/libtdr_wprobeN.so python profiles/probe_wgrad.py
hipGraph of 20 launches alternating between two operand pairs; the split-K reduction kernel is part of each launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)


def t(name, N, Cin, Cout, H, KH=1, gate=False):
    xs = [torch.randn(N, Cin * (2 if gate else 1), H, H, device='cuda') for _ in range(2)]
    ds = [torch.randn(N, Cout, H, H, device='cuda') for _ in range(2)]
    f = lambda i: K.conv_wgrad(xs[i & 1], ds[i & 1], Cout, Cin, KH, pad=KH // 2, want_db=True, fp16_range=True, gate=gate)
    for i in range(3): f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(20): f(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:30s} {e0.elapsed_time(e1) / 100 * 1e3:8.1f} us', flush=True)


print(os.environ.get('TDR_LIB_PATH', 'product'), 'TDR_WG_D1 =', os.environ.get('TDR_WG_D1', '1'))
t('1x1 256->512 @64 N4', 4, 256, 512, 64)
t('1x1 256->256 @64 N4', 4, 256, 256, 64)
t('1x1 gated 256->256 @64 N4', 4, 256, 256, 64, gate=True)
t('1x1 512->512 @32 N4', 4, 512, 512, 32)
t('1x1 96->288 @128 N8 (Restormer)', 8, 96, 288, 128)
t('1x1 128->256 @128 N4', 4, 128, 256, 128)
t('1x1 64->128 @256 N4', 4, 64, 128, 256)
t('3x3 128->128 @128 N8', 8, 128, 128, 128, 3)
t('3x3 512->512 @32 N8', 8, 512, 512, 32, 3)

"""Ablation timing of conv_bx3_kernel under the 2-way fp16 split (profiling builds libtdr_probeN.so, TDR_PROBE 1-4, see
csrc/tdr_conv_bx3.hip).  usage: TDR_LIB_PATH=textualdegremoval_amd/libtdr_probeN.so python profiles/probe_conv_hx2.py
Each shape: hipGraph of 20 launches alternating between two input / output buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)


def t(name, N, Cin, Cout, H, KH):
    xs = [torch.randn(N, Cin, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(Cout, Cin, KH, KH, device='cuda') * 0.05
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    outs = [torch.empty(N, Cout, H, H, device='cuda') for _ in range(2)]
    f = lambda i: K.conv_forward(xs[i & 1], wp, mp, Cout, KH, pad=KH // 2, out=outs[i & 1])
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


print(os.environ.get('TDR_LIB_PATH', 'product'))
t('3x3 32->32 @512 N8', 8, 32, 32, 512, 3)
t('3x3 64->64 @256 N8', 8, 64, 64, 256, 3)
t('3x3 128->128 @128 N8', 8, 128, 128, 128, 3)
t('3x3 256->256 @64 N8', 8, 256, 256, 64, 3)
t('3x3 512->512 @32 N8', 8, 512, 512, 32, 3)
t('1x1 256->512 @128 N4', 4, 256, 512, 128, 1)
t('1x1 128->256 @256 N4', 4, 128, 256, 256, 1)

"""Ablation timing of conv1x1_hx2_kernel (profiling builds libtdr_probeNN.so, see csrc/tdr_conv_bx3.hip TDR_PROBE 11-15).
usage: TDR_LIB_PATH=textualdegremoval_amd/libtdr_probeNN.so python profiles/probe_conv1x1.py
Each shape is timed inside a hipGraph of 40 back-to-back launches on two alternating input buffers (cold-ish L2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)


def t(name, N, Cin, Cout, H):
    xs = [torch.randn(N, Cin, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    outs = [torch.empty(N, Cout, H, H, device='cuda') for _ in range(2)]
    f = lambda i: K.conv_forward(xs[i & 1], wp, mp, Cout, 1, pad=0, out=outs[i & 1])
    for i in range(3): f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(40): f(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:28s} {e0.elapsed_time(e1) / 200 * 1e3:8.1f} us', flush=True)


print(os.environ.get('TDR_LIB_PATH', 'product'), 'TDR_C1_OLD' in os.environ and 'old kernel' or '')
t('1x1 256->512 @64 N4', 4, 256, 512, 64)
t('1x1 256->256 @64 N4', 4, 256, 256, 64)
t('1x1 512->256 @64 N4', 4, 512, 256, 64)
t('1x1 64->128 @512 N4', 4, 64, 128, 512)
t('1x1 128->256 @256 N4', 4, 128, 256, 256)
t('1x1 32->64 @512 N4', 4, 32, 64, 512)

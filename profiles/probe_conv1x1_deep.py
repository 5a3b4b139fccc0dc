"""1x1 convolutions of the C = 512 / 1024 levels (3 NAFBlocks at 32 x 32, N = 4: 4096 pixels, up to 2048 x 1024 weights) on conv_bx3_kernel under
the default arithmetic, per forced tile configuration (tdr_conv_force_cfg(1, cfg): 0 heuristic, 1 128x128, 2 64x256, 3 64x128, 4 32x256, 5 256x64;
co x pixels).  A launch streams the whole weight pack once per pixel tile: (pixels / tile) x |W| bytes from L2 / MALL.
usage: python profiles/probe_conv1x1_deep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
K.set_math('bx3')
torch.manual_seed(0)
lib = _lib.load()


def t(N, Cin, Cout, H, mode):
    xs = [torch.randn(N, Cin, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
    wp, mp, *_ = K.pack_weights(w, mode)
    co = Cout if mode == K.PACK_FWD else Cin
    xin = xs if mode == K.PACK_FWD else [torch.randn(N, Cout, H, H, device='cuda') for _ in range(2)]
    outs = [torch.empty(N, co, H, H, device='cuda') for _ in range(2)]
    line = f'1x1 {Cin}->{Cout} @{H} N{N} {"fwd  " if mode == K.PACK_FWD else "dgrad"}:'
    flop = 2.0 * N * Cin * Cout * H * H
    for cfg in (0, 1, 2, 3, 4, 5):
        lib.tdr_conv_force_cfg(1, cfg)
        f = lambda i: K.conv_forward(xin[i & 1], wp, mp, co, 1, pad=0, out=outs[i & 1])
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
        us = e0.elapsed_time(e1) / 100 * 1e3
        line += f'  c{cfg} {us:6.1f} us ({flop / us * 1e-6:4.0f} TF)'
    lib.tdr_conv_force_cfg(1, 0)
    print(line, flush=True)


for (Cin, Cout) in [(1024, 2048), (1024, 1024), (1024, 512), (512, 1024), (512, 512)]:
    t(4, Cin, Cout, 32, K.PACK_FWD)
    t(4, Cin, Cout, 32, K.PACK_DGRAD_S1)

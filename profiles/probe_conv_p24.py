"""Timing of the triple-plane 3x3 convolution (csrc/tdr_conv_p16.hip, PF_TRI) at the MASA-encoder levels of configs[1] (N = 8 stacked
images) for every tile configuration, next to conv_bx3_kernel<SCH_BX3> on fp32 tensors, and of the triple-plane weight gradient.
usage: python profiles/probe_conv_p24.py [cfg,cfg,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
K.set_math('bx3')
torch.manual_seed(0)
lib = _lib.load()
CFGS = [int(c) for c in sys.argv[1].split(',')] if len(sys.argv) > 1 else [301, 302, 303, 304, 306, 307, 311, 312, 321]


def bench(fn, reps=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def level(N, Cc, H):
    xs = [torch.randn(N, Cc, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(Cc, Cc, 3, 3, device='cuda') * 0.05
    b = torch.randn(Cc, device='cuda')
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    outs = [torch.empty(N, Cc, H, H, device='cuda') for _ in range(2)]
    t_old = bench(lambda i: K.conv_forward(xs[i & 1], wp, mp, Cc, 3, pad=1, bias=b, relu=True, out=outs[i & 1]))
    x3 = [K.p16_from_f32(x, fmt=K.FMT_BX3) for x in xs]
    flop = 2.0 * N * Cc * Cc * 9 * H * H
    line = f'3x3 {Cc}->{Cc} @{H} N{N}: fp32-tensor {t_old:6.1f} us ({flop / t_old * 1e-6:4.0f} TF) |'
    for cfg in CFGS:
        lib.tdr_conv3x3_p16_force_cfg(cfg)
        try:
            t = bench(lambda i: K.conv3x3_p16(x3[i & 1], wp, mp, Cc, bias=b, relu=True, want32=False, want16=True))
            line += f' c{cfg} {t:6.1f}'
        except Exception as e:
            line += f' c{cfg} ERR'
    lib.tdr_conv3x3_p16_force_cfg(0)
    d3 = x3[1]
    t_wg = bench(lambda i: K.wgrad3x3_p16(x3[0], d3, want_db=True))
    t_wg_old = bench(lambda i: K.conv_wgrad(xs[0], xs[1], Cc, Cc, 3, pad=1, want_db=True))
    line += f' | wgrad planes {t_wg:6.1f} us, fp32-tensor {t_wg_old:6.1f} us'
    print(line, flush=True)


if __name__ == '__main__':
    for Cc, H in (((32, 512),) if os.environ.get('PROBE_C32') else ((64, 256), (128, 128), (256, 64), (512, 32))):
        level(8, Cc, H)

"""Thin triple-plane 3x3 convolution (conv3x3_p24_thin_kernel, csrc/tdr_conv_p16.hip: weights in registers, persistent 4-wave workgroups)
against the general kernel at the C = 32 level: bit-identity over the epilogue variants and ragged shapes, then time per launch at
32 -> 32 @ 512^2, N = 8.  usage: python profiles/probe_p24_thin.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
K.set_math('bx3')
torch.manual_seed(0)
lib = _lib.load()


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


def run(cfg, x3, wp, mp, Cc, **kw):
    lib.tdr_conv3x3_p16_force_cfg(cfg)
    try:
        return K.conv3x3_p16(x3, wp, mp, Cc, **kw)
    finally:
        lib.tdr_conv3x3_p16_force_cfg(0)


bad = 0
worst = 0.0
for (N, Cin, Cout, H, W) in [(2, 32, 32, 64, 64), (1, 32, 32, 40, 72), (3, 32, 32, 13, 50), (2, 16, 32, 24, 32), (2, 32, 16, 36, 96)]:
    x = torch.randn(N, Cin, H, W, device='cuda'); r = torch.randn(N, Cout, H, W, device='cuda'); m = torch.randn(N, Cout, H, W, device='cuda').relu()
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05; b = torch.randn(Cout, device='cuda')
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    x3, r3, m3 = (K.p16_from_f32(t, fmt=K.FMT_BX3) for t in (x, r, m))
    for name, kw in [('bias+relu->planes', dict(bias=b, relu=True, want32=False, want16=True)),
                     ('bias+res16->f32+planes', dict(bias=b, res=r3, want32=True, want16=True)),
                     ('mask16->planes', dict(mask=m3, want32=False, want16=True)),
                     ('res32+mask32->f32', dict(res=r, mask=m, want32=True, want16=False))]:
        ref32, ref16 = run(308, x3, wp, mp, Cout, **kw)
        for cfg in (330,):
            o32, o16 = run(cfg, x3, wp, mp, Cout, **kw)
            d32 = 0.0 if ref32 is None else (o32 - ref32).abs().max().item() / max(ref32.abs().max().item(), 1e-30)
            d16 = 0.0 if ref16 is None else (o16.to_f32() - ref16.to_f32()).abs().max().item() / max(ref16.to_f32().abs().max().item(), 1e-30)
            worst = max(worst, d32, d16)
            ok = d32 < 2e-6 and d16 < 2e-6
            if not ok:
                bad += 1
                d = (o32 - ref32).abs().max().item() if ref32 is not None else (o16.to_f32() - ref16.to_f32()).abs().max().item()
                print(f'MISMATCH cfg {cfg} N{N} {Cin}->{Cout} @{H}x{W} {name}: max diff {d:.3e}')
print(f'against the general kernel (another summation order over the 18 steps): worst relative difference {worst:.2e};', 'OK' if bad == 0 else f'{bad} mismatches', flush=True)

N, Cc, H = 8, 32, 512
xs = [torch.randn(N, Cc, H, H, device='cuda') for _ in range(2)]
w = torch.randn(Cc, Cc, 3, 3, device='cuda') * 0.05; b = torch.randn(Cc, device='cuda')
wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
x3 = [K.p16_from_f32(x, fmt=K.FMT_BX3) for x in xs]
flop = 2.0 * N * Cc * Cc * 9 * H * H
for label, kw in [('conv1 (bias, relu -> planes)', dict(bias=b, relu=True, want32=False, want16=True)),
                  ('conv2 (bias, + planes -> planes)', dict(bias=b, res=x3[1], want32=False, want16=True)),
                  ('dgrad (mask planes -> planes)', dict(mask=x3[1], want32=False, want16=True))]:
    line = f'32->32 @512 N8 {label}:'
    for cfg in (308, 330):
        lib.tdr_conv3x3_p16_force_cfg(cfg)
        t = bench(lambda i: K.conv3x3_p16(x3[i & 1], wp, mp, Cc, **kw))
        line += f'  c{cfg} {t:6.1f} us ({flop / t * 1e-6:4.0f} TF)'
    lib.tdr_conv3x3_p16_force_cfg(0)
    print(line, flush=True)

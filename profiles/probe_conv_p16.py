"""Probe of the pre-split 3x3 convolution (csrc/tdr_conv_p16.hip) against conv_bx3_kernel<SCH_HX2> on the same operands:
correctness (bit-exact expected: same products, same accumulation order), then timing at the five MASA-encoder levels of
configs[1] (N = 8 stacked images) for every tile configuration.  usage: python profiles/probe_conv_p16.py [check|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
K.set_math('hx2')
torch.manual_seed(0)
lib = _lib.load()
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
CFGS = [int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 3, 4, 6, 7, 8, 9, 10, 12]


def check(N, Cin, Cout, H, W, cfg, res_kind=None, mask_kind=None, relu=False, bias=True):
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(Cout, device='cuda') if bias else None
    res = torch.randn(N, Cout, H, W, device='cuda') if res_kind else None
    mask = torch.randn(N, Cout, H, W, device='cuda') if mask_kind else None
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    x16 = K.p16_from_f32(x)
    # the P16 image decodes to the split value
    xr = x16.to_f32()
    d_rt = (xr - x).abs().max().item()
    res16 = K.p16_from_f32(res) if res_kind == 'p16' else None
    mask16 = K.p16_from_f32(mask) if mask_kind == 'p16' else None
    res_ref = res16.to_f32() if res16 is not None else res
    # reference: the existing kernel; its mask test is (mask > 0) on fp32, the P16 mask tests the head plane
    mask_ref = mask
    ref = K.conv_forward(x, wp, mp, Cout, 3, pad=1, bias=b, res=res_ref, mask=mask_ref, relu=relu)
    lib.tdr_conv3x3_p16_force_cfg(cfg)
    o32, o16 = K.conv3x3_p16(x16, wp, mp, Cout, bias=b, res=res16 if res16 is not None else res,
                             mask=mask16 if mask16 is not None else mask, relu=relu, want32=True, want16=(Cout % 16 == 0))
    torch.cuda.synchronize()
    d32 = (o32 - ref).abs().max().item()
    msg = f'N{N} {Cin}->{Cout} {H}x{W} cfg{cfg} res={res_kind} mask={mask_kind} relu={relu}: roundtrip {d_rt:.2e} out32-vs-old {d32:.3e}'
    ok = d32 == 0.0
    if o16 is not None:
        back = o16.to_f32()
        # the P16 output must be the split of the fp32 output
        want = K.p16_from_f32(o32).to_f32()
        d16 = (back - want).abs().max().item()
        # and its border zero: compare whole buffers (interior + border) with the converter's image of o32
        same = torch.equal(o16.buf, K.p16_from_f32(o32).buf)
        msg += f' out16 {d16:.3e} buffers-equal {same}'
        ok = ok and d16 == 0.0 and same
    print(('OK   ' if ok else 'FAIL ') + msg, flush=True)
    return ok


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


def timing(N, Cc, H):
    xs = [torch.randn(N, Cc, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(Cc, Cc, 3, 3, device='cuda') * 0.05
    b = torch.randn(Cc, device='cuda')
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    outs = [torch.empty(N, Cc, H, H, device='cuda') for _ in range(2)]
    t_old = bench(lambda i: K.conv_forward(xs[i & 1], wp, mp, Cc, 3, pad=1, bias=b, relu=True, out=outs[i & 1]))
    x16 = [K.p16_from_f32(x) for x in xs]
    t_cvt = bench(lambda i: K.p16_from_f32(xs[i & 1], out=x16[i & 1]))
    line = f'3x3 {Cc}->{Cc} @{H} N{N}: old {t_old:7.1f} us | p16_from_f32 {t_cvt:6.1f} us |'
    flop = 2.0 * N * Cc * Cc * 9 * H * H
    for cfg in CFGS:
        if cfg not in (3, 4, 6, 30, 31, 32, 33, 34, 35, 36, 130, 131, 132) and Cc < 64:
            continue
        if cfg in (30, 31, 32, 33, 34, 35, 36, 130, 131, 132) and Cc > 32:
            continue
        lib.tdr_conv3x3_p16_force_cfg(cfg)
        for tag, kw in (('f32', dict(want32=True, want16=False)), ('p16', dict(want32=False, want16=True))):
            try:
                t = bench(lambda i: K.conv3x3_p16(x16[i & 1], wp, mp, Cc, bias=b, relu=True, out32=outs[i & 1] if kw['want32'] else None, **kw))
                line += f' cfg{cfg}/{tag} {t:6.1f}'
            except Exception as e:     # noqa: BLE001
                line += f' cfg{cfg}/{tag} ERR({str(e)[:40]})'
    lib.tdr_conv3x3_p16_force_cfg(0)
    t = bench(lambda i: K.conv3x3_p16(x16[i & 1], wp, mp, Cc, bias=b, relu=True, out32=outs[i & 1], want32=True, want16=True))
    line += f' | auto/both {t:6.1f} us = {flop / t * 1e-6:6.1f} TF'
    print(line, flush=True)


if what in ('check', 'all'):
    allok = True
    for cfg in CFGS:
        allok &= check(2, 32, 64, 32, 32, cfg)
        allok &= check(1, 64, 128, 40, 64, cfg, res_kind='f32', relu=True)
        allok &= check(2, 48, 32, 19, 45, cfg, res_kind='p16', mask_kind='f32')
        allok &= check(1, 16, 16, 8, 8, cfg, mask_kind='p16', bias=False)
        allok &= check(1, 128, 128, 64, 64, cfg, res_kind='f32', mask_kind='f32')
        allok &= check(1, 32, 24, 33, 31, cfg)
        allok &= check(2, 32, 32, 40, 70, cfg, res_kind='p16', mask_kind='p16', relu=True)
        allok &= check(3, 16, 32, 17, 96, cfg, res_kind='f32')
    print('ALL OK' if allok else 'SOME FAILED', flush=True)
if what in ('time', 'all'):
    timing(8, 32, 512)
    timing(8, 64, 256)
    timing(8, 128, 128)
    timing(8, 256, 64)
    timing(8, 512, 32)

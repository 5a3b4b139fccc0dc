"""Probe of the P16 weight-gradient kernel (csrc/tdr_wgrad_p16.hip) against the fp32-input split kernel (tdr_conv_wgrad, math 2)
and an fp64 reference; then timing at the five MASA-encoder levels (N = 8).  usage: python profiles/probe_wgrad_p16.py [check|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
K.set_math('hx2')
torch.manual_seed(0)
what = sys.argv[1] if len(sys.argv) > 1 else 'all'


def ref64(x, d):
    """fp64 weight gradient of a 3x3 / pad 1 conv on the device"""
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    N, Ci, H, W = x.shape
    g = torch.empty(d.shape[1], Ci, 3, 3, dtype=torch.float64, device=x.device)
    for ky in range(3):
        for kx in range(3):
            g[:, :, ky, kx] = torch.einsum('nchw,nkhw->ck', d.double(), xp[:, :, ky:ky + H, kx:kx + W])
    return g


def check(N, Cin, Cout, H, W):
    x = torch.randn(N, Cin, H, W, device='cuda')
    d = torch.randn(N, Cout, H, W, device='cuda')
    x16, d16 = K.p16_from_f32(x), K.p16_from_f32(d)
    g, db = K.wgrad3x3_p16(x16, d16, want_db=True)
    g0, db0 = K.conv_wgrad(x, d, Cout, Cin, 3, pad=1, want_db=True, fp16_range=True)
    torch.cuda.synchronize()
    r = ref64(x16.to_f32(), d16.to_f32())
    sc = r.abs().max().item()
    e_new = (g[0].double() - r).abs().max().item() / sc
    e_old = (g0[0].double() - ref64(x, d)).abs().max().item() / sc
    e_db = (db.double() - d16.to_f32().double().sum((0, 2, 3))).abs().max().item() / max(db0.abs().max().item(), 1e-9)
    e_db0 = (db0.double() - d.double().sum((0, 2, 3))).abs().max().item() / max(db0.abs().max().item(), 1e-9)
    ok = e_new < 5e-6 and e_db < 5e-6
    print(f'{"OK  " if ok else "FAIL"} N{N} {Cin}->{Cout} {H}x{W}: new vs fp64 {e_new:.2e} (old kernel {e_old:.2e})  db {e_db:.2e} (old {e_db0:.2e})', flush=True)
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
    ds = [torch.randn(N, Cc, H, H, device='cuda') for _ in range(2)]
    K.set_grad_scaled(True)
    t_old = bench(lambda i: K.conv_wgrad(xs[i & 1], ds[i & 1], Cc, Cc, 3, pad=1, want_db=True))
    x16 = [K.p16_from_f32(x) for x in xs]
    d16 = [K.p16_from_f32(d) for d in ds]
    t_new = bench(lambda i: K.wgrad3x3_p16(x16[i & 1], d16[i & 1], want_db=True))
    flop = 2.0 * N * Cc * Cc * 9 * H * H
    print(f'wgrad3x3 {Cc}->{Cc} @{H} N{N}: old (kernel + reduce) {t_old:7.1f} us | p16 (kernel + reduce) {t_new:7.1f} us = {flop / t_new * 1e-6:6.1f} TF', flush=True)


if what in ('check', 'all'):
    allok = True
    for shape in ((1, 32, 32, 16, 32), (2, 64, 64, 32, 32), (1, 16, 48, 19, 45), (2, 128, 64, 24, 64), (1, 32, 32, 40, 33), (1, 64, 128, 7, 70)):
        allok &= check(*shape)
    print('ALL OK' if allok else 'SOME FAILED', flush=True)
if what in ('time', 'all'):
    timing(8, 32, 512)
    timing(8, 64, 256)
    timing(8, 128, 128)
    timing(8, 256, 64)
    timing(8, 512, 32)

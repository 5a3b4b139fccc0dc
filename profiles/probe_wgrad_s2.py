"""3x3 stride-2 weight gradients of the MASA encoder (conv_L2 .. conv_L5 at 8 x 512^2): the 2-way fp16 split kernel of
csrc/tdr_wgrad_s2.hip against the exact-fp32 kernel it replaces (TDR_WG_S2=0), and their agreement with a float64 reference.
usage: python profiles/probe_wgrad_s2.py   (runs itself twice, TDR_WG_S2 = 1 / 0)
hipGraph of 20 launches alternating between two operand pairs; the split-K reduction kernel is part of each launch."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if 'TDR_WG_S2' not in os.environ:
    for v in ('1', '0'):
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, TDR_WG_S2=v), check=False)
    sys.exit(0)

import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)


def t(name, N, Cin, Cout, H, KH=3):
    xs = [torch.randn(N, Cin, H, H, device='cuda') for _ in range(2)]
    ds = [torch.randn(N, Cout, H // 2, H // 2, device='cuda') for _ in range(2)]
    f = lambda i: K.conv_wgrad(xs[i & 1], ds[i & 1], Cout, Cin, KH, stride=2, pad=(KH - 1) // 2, want_db=True, fp16_range=True)
    for i in range(3): f(i)
    torch.cuda.synchronize()
    g, db = f(0)
    w = torch.zeros(Cout, Cin, KH, KH, device='cuda', dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xs[0][:2].double(), w, stride=2, padding=(KH - 1) // 2)
    g2, _ = K.conv_wgrad(xs[0][:2].contiguous(), ds[0][:2].contiguous(), Cout, Cin, KH, stride=2, pad=(KH - 1) // 2, want_db=True, fp16_range=True)
    ref = torch.autograd.grad(y, w, ds[0][:2].double())[0]
    err = ((g2.view_as(ref).double() - ref).abs().max() / ref.abs().max()).item()
    dbe = ((db.double() - ds[0].double().sum(dim=(0, 2, 3))).abs().max() / ds[0].double().sum(dim=(0, 2, 3)).abs().max()).item()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for i in range(20): f(i)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:30s} {e0.elapsed_time(e1) / 100 * 1e3:8.1f} us   rel err vs float64 (2 images) {err:.2e}   db {dbe:.2e}', flush=True)


print('TDR_WG_S2 =', os.environ['TDR_WG_S2'], ' TDR_WG_S2_WANT =', os.environ.get('TDR_WG_S2_WANT', '256'))
t('3x3 s2 32->64 @512 N8', 8, 32, 64, 512)
t('3x3 s2 64->128 @256 N8', 8, 64, 128, 256)
t('3x3 s2 128->256 @128 N8', 8, 128, 256, 128)
t('3x3 s2 256->512 @64 N8', 8, 256, 512, 64)
t('2x2 s2 32->64 @512 N4', 4, 32, 64, 512, 2)
t('2x2 s2 64->128 @256 N4', 4, 64, 128, 256, 2)
t('2x2 s2 128->256 @128 N4', 4, 128, 256, 128, 2)
t('2x2 s2 256->512 @64 N4', 4, 256, 512, 64, 2)

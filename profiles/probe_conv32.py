"""C = 32 level of the MASA encoder (fp32 tensors, conv_bx3_kernel): 3x3 32 -> 32 @ 512^2, N = 8, forward (+bias, ReLU) and the weight gradient, in the
current arithmetic.  usage: [TDR_RING3=0] [TDR_MATH=..] python profiles/probe_conv32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
from probe_conv_p24 import bench  # noqa
torch.manual_seed(0)
N, Cc, H = 8, 32, 512
xs = [torch.randn(N, Cc, H, H, device='cuda') for _ in range(2)]
w = torch.randn(Cc, Cc, 3, 3, device='cuda') * 0.05
b = torch.randn(Cc, device='cuda')
wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
outs = [torch.empty(N, Cc, H, H, device='cuda') for _ in range(2)]
t = bench(lambda i: K.conv_forward(xs[i & 1], wp, mp, Cc, 3, pad=1, bias=b, relu=True, out=outs[i & 1]))
flop = 2.0 * N * Cc * Cc * 9 * H * H
print(f'TDR_MATH={K.MATH} TDR_RING3={os.environ.get("TDR_RING3", "1")}: conv 3x3 32->32 @512 N8 {t:6.1f} us ({flop / t * 1e-6:4.0f} TF, {8 * N * Cc * H * H / t * 1e-6:5.0f} GB/s algorithmic)')
tw = bench(lambda i: K.conv_wgrad(xs[0], xs[1], Cc, Cc, 3, pad=1, want_db=True))
print(f'   wgrad 3x3 32->32 @512 N8 {tw:6.1f} us')

"""Ablation timing of conv_bx3 (profiling builds libtdr_probeN.so, see csrc/tdr_conv_bx3.hip TDR_PROBE).
usage: TDR_LIB_PATH=textualdegremoval_amd/libtdr_probeN.so python profiles/probe_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('bx3')
torch.manual_seed(0)
def t(name, N, Cin, Cout, H, KH):
    x = torch.randn(N, Cin, H, H, device='cuda'); w = torch.randn(Cout, Cin, KH, KH, device='cuda') * 0.05
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD); pad = 1 if KH == 3 else 0
    out = torch.empty(N, Cout, H, H, device='cuda')
    f = lambda: K.conv_forward(x, wp, mp, Cout, KH, pad=pad, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:28s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us', flush=True)
print(os.environ.get('TDR_LIB_PATH', 'product'))
t('3x3 L1 32->32 @512 N8', 8, 32, 32, 512, 3)
t('3x3 L3 128->128 @128 N8', 8, 128, 128, 128, 3)
t('3x3 L5 512->512 @32 N8', 8, 512, 512, 32, 3)
t('1x1 256->512 @64 N4', 4, 256, 512, 64, 1)
t('1x1 256->256 @64 N4', 4, 256, 256, 64, 1)
t('1x1 64->128 @512 N4', 4, 64, 128, 512, 1)
t('1x1 128->256 @256 N4', 4, 128, 256, 256, 1)
t('1x1 128->128 @256 N4', 4, 128, 128, 256, 1)
t('1x1 32->64 @512 N4', 4, 32, 64, 512, 1)
t('1x1 512->1024 @64 N4', 4, 512, 1024, 64, 1)

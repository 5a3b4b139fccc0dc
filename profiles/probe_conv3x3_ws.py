"""A/B of the producer/consumer 3x3 kernel (csrc/tdr_conv3x3_ws.hip) against conv_bx3_kernel on the MASA-encoder shapes:
run once with TDR_CONV_WS=1 (default) and once with TDR_CONV_WS=0; prints time per launch and a checksum of the output
(the two kernels are bit-identical).  Each shape: hipGraph of 20 launches alternating between two input / output buffers."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)


def t(name, N, C, H, res=False, relu=False, mask=False):
    xs = [torch.randn(N, C, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    b = torch.randn(C, device='cuda')
    r = torch.randn(N, C, H, H, device='cuda') if res else None
    m = torch.randn(N, C, H, H, device='cuda') if mask else None
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    outs = [torch.empty(N, C, H, H, device='cuda') for _ in range(2)]
    f = lambda i: K.conv_forward(xs[i & 1], wp, mp, C, 3, pad=1, bias=b, res=r, relu=relu, mask=m, out=outs[i & 1])
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
    h = hashlib.sha256(torch.cat([o.flatten() for o in outs]).cpu().numpy().tobytes()).hexdigest()[:12]
    print(f'{name:34s} {e0.elapsed_time(e1) / 100 * 1e3:8.1f} us  {h}', flush=True)


print('TDR_CONV_WS =', os.environ.get('TDR_CONV_WS', '1'))
t('3x3 64->64 @256 N8 relu', 8, 64, 256, relu=True)
t('3x3 128->128 @128 N8 relu', 8, 128, 128, relu=True)
t('3x3 128->128 @128 N8 res+mask', 8, 128, 128, res=True, mask=True)
t('3x3 256->256 @64 N8 res', 8, 256, 64, res=True)
t('3x3 512->512 @32 N8 res', 8, 512, 32, res=True)
t('3x3 128->128 @72 N3 (ragged)', 3, 128, 72, res=True)
t('3x3 256->256 @20 N2 (tw16)', 2, 256, 20, relu=True)

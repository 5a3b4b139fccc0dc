import os, sys
sys.path.insert(0, '/root/repo')
import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)
def t(name, N, C, H, res=False, relu=False):
    xs = [torch.randn(N, C, H, H, device='cuda') for _ in range(2)]
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    b = torch.randn(C, device='cuda')
    r = torch.randn(N, C, H, H, device='cuda') if res else None
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    outs = [torch.empty(N, C, H, H, device='cuda') for _ in range(2)]
    f = lambda i: K.conv_forward(xs[i & 1], wp, mp, C, 3, pad=1, bias=b, res=r, relu=relu, out=outs[i & 1])
    for i in range(3): f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(20): f(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:30s} {e0.elapsed_time(e1) / 100 * 1e3:8.1f} us', flush=True)
print(os.environ.get('TDR_LIB_PATH', 'product'))
t('3x3 32->32 @512 N8 relu', 8, 32, 512, relu=True)
t('3x3 32->32 @512 N8 res', 8, 32, 512, res=True)
t('3x3 64->64 @256 N8 relu', 8, 64, 256, relu=True)

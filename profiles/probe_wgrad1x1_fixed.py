"""1x1 weight gradient 256 -> 512, N = 4 at growing contraction lengths: fixed cost per launch (ramp, prologue, in-block K combine, partial
write, split-K reduction) against the cost per 32-pixel stage.  usage: [TDR_WG1_SP=0|1] python profiles/probe_wgrad1x1_fixed.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
torch.manual_seed(0)


def bench(fn, reps=10):
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


res = []
for H in (32, 64, 128, 256):
    x = torch.randn(4, 256, H, H, device='cuda')
    d = torch.randn(4, 512, H, H, device='cuda')
    t = bench(lambda i: K.conv_wgrad(x, d, 512, 256, 1, want_db=True))
    stages = 4 * H * H // 32 * 8 // 256
    res.append((stages, t))
    print(f'SP={os.environ.get("TDR_WG1_SP", "1")} @{H}: {stages:4d} stages per workgroup  {t:8.1f} us', flush=True)
(s0, t0), (s1, t1) = res[1], res[3]
per = (t1 - t0) / (s1 - s0)
print(f'   per stage {per:.2f} us, fixed (from the 16-stage launch) {t0 - 16 * per:.1f} us')

"""Do two half-batch kernel chains on two streams beat one full-batch chain?  (single-round L4 kernels serialise their
load / MFMA / store phases; two independent chains could interleave them.)  Chain = LN -> 1x1 c->2c -> dwsg -> 1x1 c->c
-> LN -> 1x1 c->2c -> gated 1x1, all at 64x64 with c = 256, captured in a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('hx2')
torch.manual_seed(0)
c, H = 256, 64
dev = 'cuda'
w1 = torch.randn(2 * c, c, 1, 1, device=dev) * 0.05; w3 = torch.randn(c, c, 1, 1, device=dev) * 0.05
w4 = torch.randn(2 * c, c, 1, 1, device=dev) * 0.05; w5 = torch.randn(c, c, 1, 1, device=dev) * 0.05
wd = torch.randn(2 * c, 1, 3, 3, device=dev) * 0.1; bd = torch.zeros(2 * c, device=dev)
lw = torch.ones(c, device=dev); lb = torch.zeros(c, device=dev)
p1, m1, *_ = K.pack_weights(w1, K.PACK_FWD); p3, m3, *_ = K.pack_weights(w3, K.PACK_FWD)
p4, m4, *_ = K.pack_weights(w4, K.PACK_FWD); p5, m5, *_ = K.pack_weights(w5, K.PACK_FWD)


def block(x):
    xn, _, _ = K.layernorm2d_fwd(x, lw, lb, 1e-6)
    t1 = K.conv_forward(xn, p1, m1, 2 * c, 1)
    g, pooled = K.dwsg_fwd(t1, wd, bd)
    y = K.conv_forward(g, p3, m3, c, 1, res=x)
    yn, _, _ = K.layernorm2d_fwd(y, lw, lb, 1e-6)
    t4 = K.conv_forward(yn, p4, m4, 2 * c, 1)
    return K.conv_forward(t4, p5, m5, c, 1, gate=True, res=y)


def chain(x, n=8):
    for _ in range(n):
        x = block(x)
    return x


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.randn(4, c, H, H, device=dev) * 0.5
xa, xb = x[:2].contiguous(), x[2:].contiguous()
chain(x, 1); chain(xa, 1); torch.cuda.synchronize()

s = torch.cuda.Stream()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g1, stream=s):
        chain(x)
t1 = timed(g1.replay)

s2 = torch.cuda.Stream()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g2, stream=s):
        ev = torch.cuda.Event(); ev.record(s)
        chain(xa)
        s2.wait_event(ev)
        with torch.cuda.stream(s2):
            chain(xb)
            ev2 = torch.cuda.Event(); ev2.record(s2)
        s.wait_event(ev2)
t2 = timed(g2.replay)
print(f'one chain, N=4: {t1 * 1e3:.1f} us   two chains of N=2 on two streams: {t2 * 1e3:.1f} us   ({8 * 7} kernels per chain)')

import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from textualdegremoval_amd.i2t import cross_attention
B, Tq, dim, heads = 4, 256, 1280, 20
g = torch.Generator().manual_seed(dim)
P = {'to_q.weight': torch.randn(dim, dim, generator=g) * dim ** -0.5, 'to_k_global.weight': torch.randn(dim, 1024, generator=g) / 32,
     'to_v_global.weight': torch.randn(dim, 1024, generator=g) / 32, 'to_out.0.weight': torch.randn(dim, dim, generator=g) * dim ** -0.5,
     'to_out.0.bias': torch.zeros(dim)}
P = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
hid = torch.randn(B, Tq, dim, device='cuda', requires_grad=True)
ctx = torch.randn(B, 77, 1024, device='cuda', requires_grad=True)
for _ in range(3):
    out = cross_attention(P, hid, ctx, heads, 64 ** -0.5); out.backward(torch.ones_like(out))
torch.cuda.synchronize()
import cProfile, pstats, time
def step():
    out = cross_attention(P, hid, ctx, heads, 64 ** -0.5); out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
t0 = time.perf_counter(); step(); print('wall ms', (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable(); step(); step(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)

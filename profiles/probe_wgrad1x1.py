"""1x1 weight gradient: wgrad1x1_dma_kernel (csrc/tdr_wgrad_1x1.hip; TDR_WG1=0: the staged kernel of rounds 1-4) -- error against fp64 and
time per launch (kernel + split-K reduction) at the NAFBlock shapes of configs[1].  usage: [TDR_WG1=0] [TDR_MATH=hx2] python profiles/probe_wgrad1x1.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
torch.manual_seed(0)
if K.MATH == 'hx2':
    K.set_grad_scaled(True)


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


print(f'TDR_MATH={K.MATH} TDR_WG1={os.environ.get("TDR_WG1", "1")} TDR_WG1_RING={os.environ.get("TDR_WG1_RING", "2")}')
for (N, Cin, Cout, H, gate, per_image) in [(4, 256, 512, 64, False, False), (4, 256, 256, 64, True, False), (4, 256, 256, 64, False, True),
                                          (4, 128, 256, 128, False, False), (4, 64, 128, 256, False, False), (4, 64, 64, 256, True, False),
                                          (4, 128, 128, 128, False, True), (4, 256, 512, 32, False, False)]:
    x = torch.randn(N, Cin * (2 if gate else 1), H, H, device='cuda')
    d = torch.randn(N, Cout, H, H, device='cuda')
    g, db = K.conv_wgrad(x, d, Cout, Cin, 1, gate=gate, per_image=per_image, want_db=True)
    xe = (x[:, :Cin] * x[:, Cin:]) if gate else x
    ref = torch.einsum('nkp,ncp->nkc', d.double().flatten(2), xe.double().flatten(2))
    if not per_image:
        ref = ref.sum(0, keepdim=True)
    err = (g.double().view_as(ref) - ref).abs().max().item() / ref.abs().max().item()
    errb = (db.double() - d.double().sum((0, 2, 3))).abs().max().item() / d.double().sum((0, 2, 3)).abs().max().item()
    t = bench(lambda i: K.conv_wgrad(x, d, Cout, Cin, 1, gate=gate, per_image=per_image, want_db=True))
    flop = 2.0 * N * Cout * Cin * H * H
    print(f'1x1 wgrad N{N} {Cin}->{Cout} @{H} gate={int(gate)} per_image={int(per_image)}: {t:7.1f} us ({flop / t * 1e-6:5.0f} TF)  rel err {err:.1e} db {errb:.1e}', flush=True)

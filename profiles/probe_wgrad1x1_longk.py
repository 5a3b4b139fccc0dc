"""1x1 weight gradient at a LONG contraction (what a grouped / stream-K launch over all deferred leaves would see): us per (workgroup, 32-pixel
stage) of wgrad1x1_dma_kernel at 1 / 2 / 3 workgroups per CU.  usage: TDR_WG1_WANT=256|512|768 [TDR_WG1_RING=3] python profiles/probe_wgrad1x1_longk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
torch.manual_seed(0)


def bench(fn, reps=5):
    for i in range(2): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


want = int(os.environ.get('TDR_WG1_WANT', '256'))
for (N, Cin, Cout, H, gate) in [(4, 256, 512, 256, False), (4, 256, 256, 256, True), (4, 64, 128, 512, False)]:
    x = torch.randn(N, Cin * (2 if gate else 1), H, H, device='cuda')
    d = torch.randn(N, Cout, H, H, device='cuda')
    t = bench(lambda i: K.conv_wgrad(x, d, Cout, Cin, 1, gate=gate, want_db=True))
    flop = 2.0 * N * Cout * Cin * H * H
    tiles = (Cout // 128 if Cout > 64 and Cin > 64 else Cout // 64) * (Cin // 128 if Cout > 64 and Cin > 64 else Cin // 64)
    stage_tiles = N * H * H // 32 * tiles
    print(f'want {want} ring {os.environ.get("TDR_WG1_RING", "2")}: N{N} {Cin}->{Cout} @{H} gate={int(gate)}: {t:8.1f} us ({flop / t * 1e-6:5.0f} TF)  '
          f'{t * min(want, 256 * 3) / stage_tiles:.2f} us per (workgroup, stage)', flush=True)

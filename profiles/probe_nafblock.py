"""Times one NAFBlock of the 64x64 level (c = 256, N = 4) forward and backward, fused halves (csrc/tdr_nafblock.hip)
against the per-op launch sequence: python profiles/probe_nafblock.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from textualdegremoval_amd import engine as E, kernels as K  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_hip_nafblock_fused import block_params, rnd  # noqa: E402

N, c, H, W = 4, 256, 64, 64
P = {k: v.cuda() for k, v in block_params(c, 3).items()}
x = rnd(N, c, H, W, seed=1).cuda()
dout = rnd(N, c, H, W, seed=2).cuda()
plan = K.PackPlan()
K.set_pack_plan(plan)
K.set_grad_scaled(True)
for fuse in (False, True, False, True):
    E.FUSE_TAIL = fuse
    for rep in range(3):
        out, saved = E.naf_fwd(x, P)
        dx, G = E.naf_bwd(dout, P, saved)
    plan.run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda._sleep(200_000_000)
    e[0].record()
    for rep in range(20):
        out, saved = E.naf_fwd(x, P)
    e[1].record()
    for rep in range(20):
        dx, G = E.naf_bwd(dout, P, saved)
    e[2].record()
    torch.cuda.synchronize()
    print(f'fuse={fuse}: fwd {e[0].elapsed_time(e[1]) / 20 * 1e3:.1f} us  bwd {e[1].elapsed_time(e[2]) / 20 * 1e3:.1f} us', flush=True)

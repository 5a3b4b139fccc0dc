"""Phase timeline of naf_tail_fwd_kernel (probe build -DNB_PROBE=8 of csrc/tdr_nafblock.hip, loaded through TDR_LIB_PATH):
cycle stamps of waves 0 and 5 of two workgroups at the phase boundaries."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: E402

from textualdegremoval_amd import _lib, engine as E, kernels as K  # noqa: E402
from test_hip_nafblock_fused import block_params, rnd  # noqa: E402

N, c, H, W = 4, 256, 64, 64
P = {k: v.cuda() for k, v in block_params(c, 3).items()}
x = rnd(N, c, H, W, seed=1).cuda()
plan = K.PackPlan()
K.set_pack_plan(plan)
for _ in range(5):
    out, saved = E.naf_fwd(x, P)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 64)()
lib.tdr_nb_probe_read(buf)
names = ['start', 'staged+barrier', 'conv3 gemm', 'LN + yn planes + barrier', 'conv4 gemm', 'gate planes + barriers', 'conv5 gemm', 'out stores issued']
for slot in range(4):
    ts = [buf[slot * 16 + k] for k in range(8)]
    print(f'slot {slot} (wg {"(0,0)" if slot < 2 else "(37,2)"} wave {0 if slot % 2 == 0 else 5}):')
    for k in range(1, 8):
        print(f'   {names[k]:32s} {ts[k] - ts[k - 1]:8d} cycles   (t = {ts[k] - ts[0]})')

"""Phase timeline of naf_tail_fwd_kernel / naf_tail_bwd_kernel at the 64x64 level (probe build: `make -C textualdegremoval_amd/csrc probe`,
loaded through TDR_LIB_PATH): s_memtime stamps (shader-clock cycles) of waves 0 and 5 of two workgroups at the phase boundaries.
    TDR_LIB_PATH=$PWD/textualdegremoval_amd/libtdr_hip_probe.so python profiles/probe_nafblock_timeline.py"""
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
dout = rnd(N, c, H, W, seed=2).cuda()
plan = K.PackPlan()
K.set_pack_plan(plan)
lib = C.CDLL(_lib.LIB_PATH)


def show(title, names):
    buf = (C.c_ulonglong * 64)()
    lib.tdr_nb_probe_read(buf)
    print(title)
    for slot in range(4):
        ts = [buf[slot * 16 + k] for k in range(len(names))]
        print(f'  wg {"(0,0)" if slot < 2 else "(37,2)"} wave {0 if slot % 2 == 0 else 5}: ' +
              '  '.join(f'{names[k]} {ts[k] - ts[k - 1]}' for k in range(1, len(names))) + f'  | total {ts[-1] - ts[0]}')


for _ in range(5):
    out, saved = E.naf_fwd(x, P)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
show('naf_tail_fwd<256> (shader cycles between phase boundaries)',
     ['start', 'loads+stage+barrier', 'conv3', 'LN+planes+barrier', 'conv4', 'gate+barriers', 'conv5', 'out stores'])
w5t, w4t = K.pack_weights(P['conv5.weight'], K.PACK_DGRAD_S1)[0], K.pack_weights(P['conv4.weight'], K.PACK_DGRAD_S1)[0]
w3t = K.pack_weights(P['conv3.weight'], K.PACK_DGRAD_S1)[0]
xs, xn, mu1, rs1, t1, g, pooled, s, y, yn, mu2, rs2, t4, c_out = saved
for i in range(6):
    if i == 3:
        e[0].record()
    K.naf_tail_bwd(dout, P['gamma'].view(-1), t4, y, mu2, rs2, P['norm2.weight'], w5t, w4t, w3tp=w3t, beta=P['beta'].view(-1), sca=s.contiguous())
e[1].record()
torch.cuda.synchronize()
print(f'(naf_tail_bwd eager, 3 launches: {e[0].elapsed_time(e[1]) / 3 * 1e3:.1f} us each incl. launch gaps)')
show('naf_tail_bwd<256, tail>',
     ['start', 'loads+stage+barrier', 'conv5T', 'gate bwd+planes+barrier', 'conv4T (two K halves)', 'LN bwd+planes+barrier', 'conv3T', 'dgp stores'])

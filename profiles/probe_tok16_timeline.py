"""s_memtime timeline of tok_gemm_kernel (apply profiles/probes/tok16_timeline_probe.patch to csrc/tdr_tok16.hip, build that object, link it with the other
objects of the library into profiles/ab/libtdr_hip_probe.so -- the product source carries no probe code):
TDR_LIB_PATH=$PWD/profiles/ab/libtdr_hip_probe.so python profiles/probe_tok16_timeline.py
Stamps of wave 0 of three workgroups: start | prologue done | per even stage: after barrier, after the global-load issue, after the
MFMA block, after the LDS stores | loop end.  s_memtime ticks at 100 MHz on gfx950 (10 ns)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
lib = _lib.load()
lib.tdr_tok_probe_read.argtypes = [C.c_void_p]
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
P = 31488
for name, N, Kd, epi in (('fc2', 768, 3072, 2), ('qkv', 2304, 768, 0)):
    x, w, bias, ls, res = r(P, Kd).half(), (r(N, Kd) * 0.03).half(), r(N), r(N), r(P, N)
    for _ in range(3):
        K.tok16_gemm(x, w, bias, epi=epi, res=res if epi == 2 else None, ls=ls if epi == 2 else None)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 192)()
    lib.tdr_tok_probe_read(buf)
    print(name)
    for blk in range(3):
        ts = [buf[blk * 64 + i] for i in range(64)]
        n = max((i for i in range(64) if ts[i]), default=-1) + 1
        d = [ts[i] - ts[i - 1] for i in range(1, n)]
        print(f'  block {blk}: {n} stamps, total {ts[n - 1] - ts[0] if n else 0} ticks; deltas {d}')

"""s_memtime timeline of conv_bx3 (TDR_PROBE=5 build): TDR_LIB_PATH=.../libtdr_probe5.so python profiles/probe_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K, _lib
K.set_math('bx3')
lib = _lib.load()
def run(name, N, Cin, Cout, H, KH):
    x = torch.randn(N, Cin, H, H, device='cuda'); w = torch.randn(Cout, Cin, KH, KH, device='cuda') * 0.05
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD); pad = 1 if KH == 3 else 0
    out = torch.empty(N, Cout, H, H, device='cuda')
    for _ in range(3): K.conv_forward(x, wp, mp, Cout, KH, pad=pad, out=out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 512)()
    lib.tdr_probe_read.argtypes = [C.c_void_p]; lib.tdr_probe_read(buf)
    print(name)
    for blk in range(3):
        ts = [buf[blk * 64 + i] for i in range(64)]
        n = max(i for i in range(64) if ts[i]) + 1 if any(ts) else 0
        d = [ts[i] - ts[i - 1] for i in range(1, n)]
        print(f'  block {blk}: total {ts[n-1]-ts[0] if n else 0} ticks; deltas {d}')
run('1x1 256->256 @64 N4', 4, 256, 256, 64, 1)
run('1x1 256->512 @64 N4', 4, 256, 512, 64, 1)
run('3x3 128->128 @128 N8', 8, 128, 128, 128, 3)
run('3x3 32->32 @512 N8', 8, 32, 32, 512, 3)

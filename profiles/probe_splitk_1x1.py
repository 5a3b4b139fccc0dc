"""Is split-K worth it for the single-round long-K 1x1 GEMMs of the ViTs (CLIP: 1280 -> 1280 / 3840 / 5120 over 1152 tokens)?
Split-K is expressible with the existing kernel: the K chunks become `images` (x [1, K, H, W] viewed as [S, K/S, H, W], per-image
packed weights via wp_ns), giving partial outputs [S, Cout, H, W].  Times the full launch against the S-way partial launch."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from textualdegremoval_amd import kernels as K
from textualdegremoval_amd.kernels import PACK_FWD

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

H, Wd = 36, 32          # 1152 tokens
for Cin, Cout in ((1280, 1280), (1280, 3840), (1280, 5120), (5120, 1280)):
    x = torch.randn(1, Cin, H, Wd, device='cuda')
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * Cin ** -0.5
    wp, mp, *_ = K.pack_weights(w, PACK_FWD)
    full = t(lambda: K.conv_forward(x, wp, mp, Cout, 1))
    row = [f'{Cin}->{Cout}: full {full:.1f} us']
    ref = K.conv_forward(x, wp, mp, Cout, 1)
    for S in (2, 4, 8):
        kc = Cin // S
        packs = [K.pack_weights(w[:, s * kc:(s + 1) * kc].contiguous(), PACK_FWD)[0] for s in range(S)]
        per = packs[0].buf.numel()
        big = torch.cat([p.buf for p in packs])
        wps = K.PackedWeights(big, packs[0].fmt)
        xs = x.view(S, kc, H, Wd)
        part = t(lambda: K.conv_forward(xs, wps, mp, Cout, 1, wp_ns=per))
        out = K.conv_forward(xs, wps, mp, Cout, 1, wp_ns=per).sum(0, keepdim=True)
        err = (out - ref).abs().max().item()
        row.append(f'S={S}: partial launch {part:.1f} us (err {err:.1e})')
    print(' | '.join(row))

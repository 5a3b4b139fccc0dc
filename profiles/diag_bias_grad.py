"""Where does the 5e-4..8e-4 (relative to the tensor maximum) gap between the product backward pass and the oracle's autograd on a few
BIAS gradients come from?  One 512x512 pair of the headline network; the gradients of the HIP path in its
exact-fp32 and fp16-split arithmetic, and of the oracle in fp32 and in fp64 (same match decisions), compared tensor by tensor against
the fp64 oracle.  Run on the GPU box:  python profiles/diag_bias_grad.py [size]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nafnet_ref_oracle as O                       # noqa: E402  (diagnostic: the oracle is the checker here)
from textualdegremoval_amd import engine as E, kernels as K     # noqa: E402

SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = O.default_cfg(width=32, nf=32, enc_blk_nums=[1, 1, 1, 28], ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])
P = O.synth_params(cfg, seed=3)
Pc = {k: v.cuda() for k, v in P.items()}
lq, gt, ref = O.synth_pair(1, SIZE, SIZE, seed=81)
S = 2.0 ** math.floor(math.log2(512.0 * 3 * SIZE * SIZE))


def hip(mode):
    prev_m = K.MATH
    K.set_math(mode)
    prev = K.set_grad_scaled(mode == 'hx2')
    try:
        out, saved = E.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
        s = S if mode == 'hx2' else 1.0
        loss, dpred = K.l1_loss(out.contiguous(), gt.cuda(), 1.0, grad_scale=s)
        G = {k: v.cpu().double() / s for k, v in E.net_bwd(dpred, Pc, cfg, saved).items()}
    finally:
        K.set_grad_scaled(prev)
        K.set_math(prev_m)
    return G, saved


G_hx2, saved = hip('hx2')
G_f32, _ = hip('f32')
sv_masa = saved[6]
hip_index, hip_index_all = sv_masa[4].cpu().long(), sv_masa[7].cpu().long()
orig_cs, orig_fs = O.coarse_search, O.fine_search


def cs(lrb, r4, dil):
    total, index = orig_cs(lrb, r4, dil)
    return total, hip_index.view_as(index)


def fs(lrb_flat, refb):
    val, idx, corr = orig_fs(lrb_flat, refb)
    hi = hip_index_all.view(corr.shape[0], -1)
    return corr.gather(2, hi.unsqueeze(2)).squeeze(2).view_as(val), hi.view_as(idx), corr


O.coarse_search, O.fine_search = cs, fs


def oracle(dt):
    Pr = {k: v.clone().to(dt).requires_grad_(True) for k, v in P.items()}
    rl = O.l1_loss(O.nafnet_ref_forward(Pr, cfg, lq.to(dt), ref.to(dt)), gt.to(dt))
    rl.backward()
    return {k: p.grad.double() for k, p in Pr.items() if p.grad is not None}


R32 = oracle(torch.float32)
R64 = oracle(torch.float64)


def rel(A, B, k):
    return (A[k].reshape(B[k].shape) - B[k]).abs().max().item() / max(B[k].abs().max().item(), 1e-300)


rows = []
for k in R64:
    rows.append((rel(G_hx2, R32, k), rel(G_hx2, R64, k), rel(G_f32, R64, k), rel(R32, R64, k), k))
rows.sort(reverse=True)
print(f'size {SIZE}: relative (to the tensor max) gaps;  hx2-vs-oracle32 | hx2-vs-oracle64 | f32-vs-oracle64 | oracle32-vs-oracle64')
for r in rows[:25]:
    print(f'  {r[0]:.2e} | {r[1]:.2e} | {r[2]:.2e} | {r[3]:.2e}   {r[4]}')
for j, name in ((1, 'hx2 vs oracle64'), (2, 'f32 vs oracle64'), (3, 'oracle32 vs oracle64')):
    w = max(rows, key=lambda r: r[j])
    print(f'worst {name}: {w[j]:.2e} at {w[4]}')

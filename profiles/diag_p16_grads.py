"""Which parameter gradients differ between the unscaled bf16-split backward and the loss-scaled backward, with and without the P16
path (tests/test_hip_full_size.py::test_full_size_loss_scaled_backward protocol)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import nafnet_ref_oracle as O
from textualdegremoval_amd import engine as E, kernels as K
SIZE = 512
cfg = O.default_cfg(width=32, nf=32, enc_blk_nums=[1, 1, 1, 28], ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])
P = O.synth_params(cfg, seed=3)
Pc = {k: v.cuda() for k, v in P.items()}
K.set_math('hx2')
lq, gt, ref = O.synth_pair(4, SIZE, SIZE, seed=80)
lq, ref, gt = lq.cuda(), ref.cuda(), gt.cuda()
S = 2.0 ** math.floor(math.log2(512.0 * lq.shape[0] * 3 * SIZE * SIZE))


def grads(gs, p16):
    E.P16_ON = p16
    prev = K.set_grad_scaled(gs != 1.0)
    try:
        out, saved = E.net_fwd(Pc, cfg, lq, ref)
        loss, dpred = K.l1_loss(out.contiguous(), gt, 1.0, grad_scale=gs)
        G = E.net_bwd(dpred, Pc, cfg, saved)
        G = {k: v.clone() / gs for k, v in G.items()}
    finally:
        K.set_grad_scaled(prev)
    return loss.item(), G


l0, G0 = grads(1.0, False)
l1, G1 = grads(S, False)
l2, G2 = grads(S, True)
print('loss', l0, l1, l2)
for name, G in (('scaled, fp32 tensors', G1), ('scaled, P16', G2)):
    rows = sorted(((G[k] - G0[k]).abs().max().item() / max(G0[k].abs().max().item(), 1e-30), k) for k in G0)
    print(name, 'worst 8:')
    for r, k in rows[-8:]:
        print(f'   {r:.3e}  {k}  max|g| {G0[k].abs().max().item():.3e}')
rows = sorted(((G2[k] - G1[k]).abs().max().item() / max(G1[k].abs().max().item(), 1e-30), k) for k in G1)
print('P16 vs fp32 tensors (both scaled), worst 8:')
for r, k in rows[-8:]:
    print(f'   {r:.3e}  {k}')

# ---- second part: inside the P16 backward, re-run every P16 launch on the fp32-tensor kernels from the DECODED operands and report
print('\nper-launch comparison inside the scaled P16 step (backward launches only):')
orig_c, orig_w = K.conv3x3_p16, K.wgrad3x3_p16
stats = []


def stat(t):
    a = t.abs()
    mx = a.max().item()
    return mx, (a < 2.0 ** -14).float().mean().item(), (a < 2.0 ** -25).float().mean().item()


def c16(x16, wp, mp, Cout, bias=None, res=None, mask=None, relu=False, want32=True, want16=False, out32=None):
    o32, o16 = orig_c(x16, wp, mp, Cout, bias=bias, res=res, mask=mask, relu=relu, want32=want32, want16=want16, out32=out32)
    if K.BACKWARD_PHASE or True:
        xin = x16.to_f32()
        r = res.to_f32() if isinstance(res, K.P16) else res
        m = mask.to_f32() if isinstance(mask, K.P16) else mask
        ref = K.conv_forward(xin, wp, mp, Cout, 3, pad=1, bias=bias, res=r, mask=m, relu=relu)
        got = o32 if o32 is not None else o16.to_f32()
        mx, f14, f25 = stat(ref)
        stats.append(('conv', x16.C, x16.H, (got - ref).abs().max().item() / max(mx, 1e-30), mx, f14, f25, stat(xin)[0]))
    return o32, o16


def w16(x16, d16, want_db=False):
    out = orig_w(x16, d16, want_db=want_db)
    g, db = out if want_db else (out, None)
    x, d = x16.to_f32(), d16.to_f32()
    g0, db0 = K.conv_wgrad(x, d, d16.C, x16.C, 3, pad=1, want_db=True)
    stats.append(('wgrad', x16.C, x16.H, (g - g0).abs().max().item() / max(g0.abs().max().item(), 1e-30),
                  (db - db0).abs().max().item() / max(db0.abs().max().item(), 1e-30) if db is not None else -1, *stat(d)))
    return out


K.conv3x3_p16, K.wgrad3x3_p16 = c16, w16
E.P16_ON = True
prev = K.set_grad_scaled(True)
out, saved = E.net_fwd(Pc, cfg, lq, ref)
loss, dpred = K.l1_loss(out.contiguous(), gt, 1.0, grad_scale=S)
stats.clear()
G = E.net_bwd(dpred, Pc, cfg, saved)
K.set_grad_scaled(prev)
for s in stats:
    if s[0] == 'conv':
        print(f'  dgrad C{s[1]:4d} H{s[2]:4d}: out vs fp32-tensor kernel {s[3]:.2e} of max {s[4]:.3e}; out elems < 2^-14: {s[5]:.3f}, < 2^-25: {s[6]:.3f}; max|in| {s[7]:.3e}')
    else:
        print(f'  wgrad C{s[1]:4d} H{s[2]:4d}: g {s[3]:.2e}  db {s[4]:.2e}; dout max {s[5]:.3e}, elems < 2^-14: {s[6]:.3f}, < 2^-25: {s[7]:.3f}')

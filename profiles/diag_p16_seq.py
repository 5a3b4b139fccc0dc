"""Sequence comparison of the backward 3x3 launches of the MASA encoder between the fp32-tensor path and the P16 path (same
loss-scaled step): first divergence."""
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
orig_c16, orig_cf, orig_w16, orig_wg = K.conv3x3_p16, K.conv_forward, K.wgrad3x3_p16, K.conv_wgrad
rec = {'p16': [], 'f32': []}
mode = [None]


def c16(x16, wp, mp, Cout, **kw):
    o32, o16 = orig_c16(x16, wp, mp, Cout, **kw)
    if K.BACKWARD_PHASE:
        rec['p16'].append(('dgrad', x16.C, x16.H, (o32 if o32 is not None else o16.to_f32()).clone(), x16.to_f32(),
                           kw.get('res').to_f32() if isinstance(kw.get('res'), K.P16) else kw.get('res'),
                           kw.get('mask').to_f32() if isinstance(kw.get('mask'), K.P16) else kw.get('mask')))
    return o32, o16


def cf(x, wp, Mpad, Cout, KH, stride=1, dil=1, pad=0, **kw):
    out = orig_cf(x, wp, Mpad, Cout, KH, stride=stride, dil=dil, pad=pad, **kw)
    if K.BACKWARD_PHASE and KH == 3 and stride == 1 and x.shape[1] >= 64 and x.shape[1] == Cout and mode[0] == 'f32' and kw.get('epi', 0) == 0:
        rec['f32'].append(('dgrad', x.shape[1], x.shape[2], out.clone(), x.clone(), kw.get('res'), kw.get('mask')))
    return out


def w16(x16, d16, **kw):
    out = orig_w16(x16, d16, **kw)
    rec['p16'].append(('wgrad', x16.C, x16.H, out[0].clone(), x16.to_f32(), d16.to_f32(), out[1].clone()))
    return out


def wg(x, dout, Cout, Cin, KH, **kw):
    out = orig_wg(x, dout, Cout, Cin, KH, **kw)
    if KH == 3 and Cin >= 64 and Cin == Cout and kw.get('stride', 1) == 1 and mode[0] == 'f32':
        rec['f32'].append(('wgrad', Cin, x.shape[2], out[0].clone(), x.clone(), dout.clone(), out[1].clone()))
    return out


K.conv3x3_p16, K.conv_forward, K.wgrad3x3_p16, K.conv_wgrad = c16, cf, w16, wg
for m in ('f32', 'p16'):
    mode[0] = m
    E.P16_ON = m == 'p16'
    prev = K.set_grad_scaled(True)
    out, saved = E.net_fwd(Pc, cfg, lq, ref)
    loss, dpred = K.l1_loss(out.contiguous(), gt, 1.0, grad_scale=S)
    G = E.net_bwd(dpred, Pc, cfg, saved)
    K.set_grad_scaled(prev)
print(len(rec['p16']), len(rec['f32']))


def rd(a, b):
    if a is None or b is None:
        return -1.0
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


for i, (a, b) in enumerate(zip(rec['p16'], rec['f32'])):
    assert a[:3] == b[:3], (a[:3], b[:3])
    if a[0] == 'dgrad':
        mflip = -1 if a[6] is None else int(((a[6] > 0) != (b[6] > 0)).sum().item())
        print(f'{i:3d} dgrad C{a[1]} H{a[2]}: out {rd(a[3], b[3]):.2e} in {rd(a[4], b[4]):.2e} res {rd(a[5], b[5]):.2e} mask flips {mflip}')
    else:
        print(f'{i:3d} wgrad C{a[1]} H{a[2]}: g {rd(a[3], b[3]):.2e} x {rd(a[4], b[4]):.2e} d {rd(a[5], b[5]):.2e} db {rd(a[6], b[6]):.2e}')

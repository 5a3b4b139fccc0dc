"""CPU ORACLE (test infrastructure, NOT product code).

Functional torch-CPU fp32 restatement of the reference's guided-restoration hot
path (NAFNet-ref train step).  Every function cites the reference file:line it
restates (paths relative to the reference checkout).  It is written as pure
functions over a parameter dict keyed by the reference's state-dict names; it
shares no code with the reference (the MASA transfer is restated in the fused
gather/overlap-average form, the searches as centre-tap correlations).

Pinned against the reference itself: tests/golden/*.npz are produced by
tests/golden/make_golden.py, which imports the reference in the build
container; tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module -- as the checker / reported baseline, never as the shipped path.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# a1  LayerNorm2d            models/archs/nafnet_arch_utils.py:264-300
# --------------------------------------------------------------------------

def layernorm2d(x, weight, bias, eps=1e-6):
    """Per-pixel normalisation over C (biased variance), affine.  The custom
    backward of the reference (:277-289) is the analytic gradient of exactly
    this expression, so plain autograd reproduces it."""
    mu = x.mean(dim=1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=1, keepdim=True)
    y = xc / torch.sqrt(var + eps)
    return y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


# --------------------------------------------------------------------------
# a2-a4  SimpleGate / NAFBlock / NAFResFuseBlock
#        models/archs/network_nafnet_guided_arch.py:170-302
# --------------------------------------------------------------------------

def simple_gate(x):
    c = x.shape[1] // 2
    return x[:, :c] * x[:, c:]


def naf_block(x, P, pre):
    """NAFBlock.forward (:216-238); NAFResFuseBlock (:279-302) is the same math."""
    c2 = P[pre + 'conv2.weight'].shape[0]
    t = layernorm2d(x, P[pre + 'norm1.weight'], P[pre + 'norm1.bias'])
    t = F.conv2d(t, P[pre + 'conv1.weight'], P[pre + 'conv1.bias'])
    t = F.conv2d(t, P[pre + 'conv2.weight'], P[pre + 'conv2.bias'], padding=1, groups=c2)
    t = simple_gate(t)
    pooled = t.mean(dim=(2, 3), keepdim=True)
    s = F.conv2d(pooled, P[pre + 'sca.1.weight'], P[pre + 'sca.1.bias'])
    t = t * s
    t = F.conv2d(t, P[pre + 'conv3.weight'], P[pre + 'conv3.bias'])
    y = x + t * P[pre + 'beta']
    t = layernorm2d(y, P[pre + 'norm2.weight'], P[pre + 'norm2.bias'])
    t = F.conv2d(t, P[pre + 'conv4.weight'], P[pre + 'conv4.bias'])
    t = simple_gate(t)
    t = F.conv2d(t, P[pre + 'conv5.weight'], P[pre + 'conv5.bias'])
    return y + t * P[pre + 'gamma']


def naf_sequence(x, P, pre, n):
    for i in range(n):
        x = naf_block(x, P, f'{pre}{i}.')
    return x


# --------------------------------------------------------------------------
# a5  ResidualBlock / Encoder  (:44-59, :110-143)
# --------------------------------------------------------------------------

def residual_block(x, P, pre):
    h = F.relu(F.conv2d(x, P[pre + 'conv1.weight'], P[pre + 'conv1.bias'], padding=1))
    return F.conv2d(h, P[pre + 'conv2.weight'], P[pre + 'conv2.bias'], padding=1) + x


def masa_encoder(x, P, pre, ext_n_blocks, levels=5):
    """Encoder.forward (:136-143).  Levels 3,4,5 all use n_blks[2] (:123-129).  `levels` < 5 stops
    early (Restormer-ref only consumes L1..L4, oracle/restormer_ref_oracle.py)."""
    counts = [ext_n_blocks[0], ext_n_blocks[1], ext_n_blocks[2], ext_n_blocks[2], ext_n_blocks[2]]
    feats = []
    for lvl in range(levels):
        k = lvl + 1
        stride = 1 if lvl == 0 else 2
        x = F.relu(F.conv2d(x, P[f'{pre}conv_L{k}.weight'], P[f'{pre}conv_L{k}.bias'],
                            stride=stride, padding=1))
        for i in range(counts[lvl]):
            x = residual_block(x, P, f'{pre}blk_L{k}.{i}.')
        feats.append(x)
    return feats


# --------------------------------------------------------------------------
# a7  check_image_size (:576-585)
# --------------------------------------------------------------------------

def pad_to_multiple(x, mult):
    h, w = x.shape[-2:]
    ph = (mult - h % mult) % mult
    pw = (mult - w % mult) % mult
    return F.pad(x, (0, pw, 0, ph))


# --------------------------------------------------------------------------
# a8  coarse search (:515-536) -- centre-tap restatement
# --------------------------------------------------------------------------

def _l2n(x, dim):
    return x / x.norm(dim=dim, keepdim=True).clamp_min(1e-12)


def lr_blocks(feat, py, px, ky, kx):
    """Replicate-pad by 1 and cut py*px overlapping (ky+2)x(kx+2) blocks
    (:627-629).  Returns [N, py*px, C, ky+2, kx+2]."""
    N, C, H, W = feat.shape
    fp = F.pad(feat, (1, 1, 1, 1), mode='replicate')
    blocks = []
    for by in range(py):
        for bx in range(px):
            blocks.append(fp[:, :, by * ky: by * ky + ky + 2, bx * kx: bx * kx + kx + 2])
    return torch.stack(blocks, dim=1)


def coarse_search(lrb, ref, dilations):
    """search (:515-536): for each dilation d correlate the 3x3 centre taps of
    every LR block (rows/cols c-d, c, c+d with c=(k+2)//2) against the d-dilated
    zero-padded 3x3 neighbourhood of every ref position; cosine similarity;
    sum over dilations; arg-max.  Returns (corr_sum [N,P,Hr*Wr], index [N,P])."""
    N, Pn, C, Kh, Kw = lrb.shape
    _, _, Hr, Wr = ref.shape
    cy, cx = Kh // 2, Kw // 2
    total = 0
    for d in dilations:
        q = lrb[:, :, :, [cy - d, cy, cy + d]][:, :, :, :, [cx - d, cx, cx + d]]
        q = _l2n(q.reshape(N, Pn, C * 9), 2)
        rp = F.pad(ref, (d, d, d, d))
        taps = [rp[:, :, ky * d: ky * d + Hr, kx * d: kx * d + Wr]
                for ky in range(3) for kx in range(3)]
        k = torch.stack(taps, dim=2).reshape(N, C * 9, Hr * Wr)   # (C,ky,kx) order
        k = _l2n(k, 1)
        total = total + torch.bmm(q, k)
    return total, total.argmax(dim=2)


# --------------------------------------------------------------------------
# a9  box arithmetic (:635-657) + grid gather with python-style index wrap
#     (:557-574, :665-678)
# --------------------------------------------------------------------------

def box_start(idx, size, diameter):
    """Start of the (diameter+2)-wide window around idx, clamped to [0,size-1]
    keeping its width; goes negative when size < diameter+2 (reference quirk,
    the gather then wraps python-style)."""
    lo = idx - diameter // 2 - 1
    hi = idx + diameter // 2 + 1
    neg = lo < 0
    lo = torch.where(neg, torch.zeros_like(lo), lo)
    hi = torch.where(neg, torch.full_like(hi, diameter + 1), hi)
    over = hi > size - 1
    hi = torch.where(over, torch.full_like(hi, size - 1), hi)
    lo = torch.where(over, hi - (diameter + 1), lo)
    return lo


def gather_ref_block(feat, y1, x1, side, s):
    """feat [N,C,H,W]; y1,x1 [N,P] block starts at the coarsest scale.
    Returns [N*P, C, side*s, side*s] with python negative-index wrap."""
    N, C, H, W = feat.shape
    Pn = y1.shape[1]
    ar = torch.arange(side * s)
    ys = (y1.reshape(N, Pn, 1) * s + ar) % H          # python-style wrap of negatives
    xs = (x1.reshape(N, Pn, 1) * s + ar) % W
    nb = torch.arange(N).view(N, 1, 1, 1)
    out = feat[nb, :, ys[:, :, :, None], xs[:, :, None, :]]      # [N,P,side*s,side*s,C]
    return out.permute(0, 1, 4, 2, 3).reshape(N * Pn, C, side * s, side * s)


# --------------------------------------------------------------------------
# a10  fine search (:495-513)
# --------------------------------------------------------------------------

def fine_search(lrb, refb):
    """search_org: lrb [B,C,k+2,k+2], refb [B,C,D,D] -> (soft_att [B,1,k,k],
    index_all [B,k,k]); cosine similarity of all 3x3 patches; the max VALUE is
    differentiable w.r.t. both inputs."""
    B, C, Kh, Kw = lrb.shape
    Dh, Dw = refb.shape[-2:]
    q = F.unfold(lrb, 3).transpose(1, 2)             # [B, k*k, C*9]
    k = F.unfold(refb, 3)                            # [B, C*9, (D-2)^2]
    corr = torch.bmm(_l2n(q, 2), _l2n(k, 1))
    val, idx = corr.max(dim=2)
    return val.view(B, 1, Kh - 2, Kw - 2), idx.view(B, Kh - 2, Kw - 2), corr


# --------------------------------------------------------------------------
# a11  transfer (:483-493, :538-555) -- fused gather / overlap-average form
# --------------------------------------------------------------------------

def bilinear_up(att, s):
    """F.interpolate(mode='bilinear', align_corners=False) by integer factor s,
    written out: src=(dst+0.5)/s-0.5 clamped at 0; right neighbour clamped."""
    B, _, h, w = att.shape

    def axis(n):
        dst = torch.arange(n * s, dtype=torch.float32)
        src = ((dst + 0.5) / s - 0.5).clamp_min(0.0)
        i0 = src.floor().long().clamp_max(n - 1)
        i1 = (i0 + 1).clamp_max(n - 1)
        lam = src - i0.float()
        return i0, i1, lam
    y0, y1, ly = axis(h)
    x0, x1, lx = axis(w)
    a = att[:, 0]
    top = a[:, y0][:, :, x0] * (1 - lx) + a[:, y0][:, :, x1] * lx
    bot = a[:, y1][:, :, x0] * (1 - lx) + a[:, y1][:, :, x1] * lx
    return (top * (1 - ly).view(1, -1, 1) + bot * ly.view(1, -1, 1)).unsqueeze(1)


def transfer(fea, index, soft_att, s, side_minus2):
    """out[b,c,Y,X] = att_up(Y,X)/cnt(Y,X) * sum over LR patches (i,j) whose
    3s x 3s footprint [i*s-s, i*s+2s) covers (Y,X) of
    fea[b,c, ry(i,j)*s + Y-(i*s-s), rx(i,j)*s + X-(j*s-s)],
    (ry,rx)=divmod(index[b,i,j], side_minus2)."""
    B, C, Hf, Wf = fea.shape
    _, Hi, Wi = index.shape
    OH, OW = Hi * s, Wi * s
    Y = torch.arange(OH)
    X = torch.arange(OW)
    acc = torch.zeros(B, C, OH, OW, dtype=fea.dtype)
    cnt = torch.zeros(OH, OW, dtype=fea.dtype)
    bidx = torch.arange(B).view(B, 1, 1)
    for di in (-1, 0, 1):
        i = Y // s + di
        vi = (i >= 0) & (i < Hi)
        ic = i.clamp(0, Hi - 1)
        for dj in (-1, 0, 1):
            j = X // s + dj
            vj = (j >= 0) & (j < Wi)
            jc = j.clamp(0, Wi - 1)
            idx = index[:, ic][:, :, jc]                       # [B,OH,OW]
            ry = idx // side_minus2
            rx = idx % side_minus2
            sy = ry * s + (Y - ic * s + s).view(1, -1, 1)
            sx = rx * s + (X - jc * s + s).view(1, 1, -1)
            valid = (vi.view(-1, 1) & vj.view(1, -1)).to(fea.dtype)
            sy = sy.clamp(0, Hf - 1)
            sx = sx.clamp(0, Wf - 1)
            g = fea[bidx, :, sy, sx].permute(0, 3, 1, 2)       # [B,C,OH,OW]
            acc = acc + g * valid
            cnt = cnt + valid
    return acc / cnt * bilinear_up(soft_att, s)


# --------------------------------------------------------------------------
# a6/a12  NAFNetRefFusion forward (:587-740)
# --------------------------------------------------------------------------

def default_cfg(**kw):
    cfg = dict(img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 1],
               dec_blk_nums=[1, 1, 1, 1], nf=16, ext_n_blocks=[4, 4, 4, 4],
               reffusion_n_blocks=[2, 2, 2, 2, 2], lr_block_size=8,
               ref_down_block_size=1.5, dilations=[1, 2, 3], psize=3)
    cfg.update(kw)
    return cfg


def masa_match_and_transfer(feat_lq, feat_ref, cfg, h, w, hr, wr, return_aux=False, padder=None):
    """:597-707.  feat_* are the pyramids (finest first; 5 levels for NAFNet-ref, 4 for
    Restormer-ref whose padder_size is 8); returns warp_ref_l (finest first, like the reference list)."""
    L = len(feat_lq)
    if padder is None:
        padder = 2 ** len(cfg['enc_blk_nums'])
    lbs = cfg['lr_block_size']
    px = w // padder // lbs
    py = h // padder // lbs
    kx = w // padder // px
    ky = h // padder // py
    dia_x = 2 * int(wr // padder // (2 * px) * cfg['ref_down_block_size']) + 1
    dia_y = 2 * int(hr // padder // (2 * py) * cfg['ref_down_block_size']) + 1
    deep_lq, deep_ref = feat_lq[L - 1], feat_ref[L - 1]
    N, C, H, W = deep_lq.shape
    Hr, Wr = deep_ref.shape[-2:]
    lrb = lr_blocks(deep_lq, py, px, ky, kx)
    with torch.no_grad():
        corr_sum, index = coarse_search(lrb, deep_ref, cfg['dilations'])
    x1 = box_start(index % Wr, Wr, dia_x)
    y1 = box_start(index // Wr, Hr, dia_y)
    side_x, side_y = dia_x + 2, dia_y + 2
    assert side_x == side_y, 'reference only works for square geometry (:668-669)'
    refb = gather_ref_block(deep_ref, y1, x1, side_x, 1)
    lrb_flat = lrb.reshape(N * py * px, C, ky + 2, kx + 2)
    soft_att, index_all, corr_fine = fine_search(lrb_flat, refb)
    warp = []
    for lvl in range(L):                       # lvl 0 = finest (scale 2^(L-1))
        s = 2 ** (L - 1 - lvl)
        blk = gather_ref_block(feat_ref[lvl], y1, x1, side_x, s)
        t = transfer(blk, index_all, soft_att, s, side_x - 2)
        Cs = t.shape[1]
        t = t.view(N, py, px, Cs, ky * s, kx * s).permute(0, 3, 1, 4, 2, 5)
        warp.append(t.reshape(N, Cs, H * s, W * s))
    if return_aux:
        aux = dict(index=index, corr_sum=corr_sum, x1=x1, y1=y1, index_all=index_all,
                   soft_att=soft_att, corr_fine=corr_fine, py=py, px=px, ky=ky, kx=kx,
                   diameter=dia_x)
        return warp, aux
    return warp


def nafnet_ref_forward(P, cfg, inp, ref, return_aux=False):
    """NAFNetRefFusion.forward(inp, ref) (:587-740)."""
    n_enc = len(cfg['enc_blk_nums'])
    H0, W0 = inp.shape[-2:]
    mult = (2 ** n_enc) * cfg['lr_block_size']
    inp = pad_to_multiple(inp, mult)
    ref = pad_to_multiple(ref, mult)
    h, w = inp.shape[-2:]
    hr, wr = ref.shape[-2:]
    feat_lq = masa_encoder(inp, P, 'masa_enc.', cfg['ext_n_blocks'])
    feat_ref = masa_encoder(ref, P, 'masa_enc.', cfg['ext_n_blocks'])
    res = masa_match_and_transfer(feat_lq, feat_ref, cfg, h, w, hr, wr, return_aux)
    warp, aux = res if return_aux else (res, None)

    x = F.conv2d(inp, P['intro.weight'], P['intro.bias'], padding=1)
    chan = x.shape[1]
    skips = []
    for lvl in range(n_enc):
        x = naf_sequence(torch.cat([x, warp[lvl]], dim=1), P, f'masa_blk_enc.{lvl}.',
                         cfg['reffusion_n_blocks'][lvl])[:, :chan]
        x = naf_sequence(x, P, f'encoders.{lvl}.', cfg['enc_blk_nums'][lvl])
        skips.append(x)
        x = F.conv2d(x, P[f'downs.{lvl}.weight'], P[f'downs.{lvl}.bias'], stride=2)
        chan *= 2
    x = naf_sequence(torch.cat([x, warp[n_enc]], dim=1), P, 'masa_blk_middle.0.',
                     cfg['reffusion_n_blocks'][n_enc])[:, :chan]
    x = naf_sequence(x, P, 'middle_blks.', cfg['middle_blk_num'])
    for lvl in range(len(cfg['dec_blk_nums'])):
        x = F.pixel_shuffle(F.conv2d(x, P[f'ups.{lvl}.0.weight']), 2)
        x = x + skips[-1 - lvl]
        x = naf_sequence(x, P, f'decoders.{lvl}.', cfg['dec_blk_nums'][lvl])
    x = F.conv2d(x, P['ending.weight'], P['ending.bias'], padding=1) + inp
    out = x[:, :, :H0, :W0]
    if return_aux:
        aux['warp'] = warp
        aux['feat_lq'] = feat_lq
        aux['feat_ref'] = feat_ref
        return out, aux
    return out


# --------------------------------------------------------------------------
# parameter construction (shapes/names = reference registration order, App. A)
# --------------------------------------------------------------------------

def param_shapes(cfg):
    """OrderedDict name -> shape in the reference's registration order
    (network_nafnet_guided_arch.py:422-479)."""
    S = OrderedDict()
    nf, width, ic = cfg['nf'], cfg['width'], cfg['img_channel']
    ext = cfg['ext_n_blocks']
    cnt = [ext[0], ext[1], ext[2], ext[2], ext[2]]
    cin = ic
    for k in range(1, 6):
        c = nf * 2 ** (k - 1)
        S[f'masa_enc.conv_L{k}.weight'] = (c, cin, 3, 3)
        S[f'masa_enc.conv_L{k}.bias'] = (c,)
        for i in range(cnt[k - 1]):
            for j in (1, 2):
                S[f'masa_enc.blk_L{k}.{i}.conv{j}.weight'] = (c, c, 3, 3)
                S[f'masa_enc.blk_L{k}.{i}.conv{j}.bias'] = (c,)
        cin = c

    def naf(pre, c):
        S[pre + 'beta'] = (1, c, 1, 1)
        S[pre + 'gamma'] = (1, c, 1, 1)
        S[pre + 'conv1.weight'] = (2 * c, c, 1, 1); S[pre + 'conv1.bias'] = (2 * c,)
        S[pre + 'conv2.weight'] = (2 * c, 1, 3, 3); S[pre + 'conv2.bias'] = (2 * c,)
        S[pre + 'conv3.weight'] = (c, c, 1, 1); S[pre + 'conv3.bias'] = (c,)
        S[pre + 'sca.1.weight'] = (c, c, 1, 1); S[pre + 'sca.1.bias'] = (c,)
        S[pre + 'conv4.weight'] = (2 * c, c, 1, 1); S[pre + 'conv4.bias'] = (2 * c,)
        S[pre + 'conv5.weight'] = (c, c, 1, 1); S[pre + 'conv5.bias'] = (c,)
        S[pre + 'norm1.weight'] = (c,); S[pre + 'norm1.bias'] = (c,)
        S[pre + 'norm2.weight'] = (c,); S[pre + 'norm2.bias'] = (c,)

    n_enc = len(cfg['enc_blk_nums'])
    chan = width
    for lvl in range(n_enc):
        for i in range(cfg['reffusion_n_blocks'][lvl]):
            naf(f'masa_blk_enc.{lvl}.{i}.', 2 * chan)
        chan *= 2
    for i in range(cfg['reffusion_n_blocks'][n_enc]):
        naf(f'masa_blk_middle.0.{i}.', 2 * chan)
    S['intro.weight'] = (width, ic, 3, 3); S['intro.bias'] = (width,)
    S['ending.weight'] = (ic, width, 3, 3); S['ending.bias'] = (ic,)
    chan = width
    for lvl in range(n_enc):
        for i in range(cfg['enc_blk_nums'][lvl]):
            naf(f'encoders.{lvl}.{i}.', chan)
        chan *= 2
    dchan = chan
    for lvl in range(len(cfg['dec_blk_nums'])):
        dchan //= 2
        for i in range(cfg['dec_blk_nums'][lvl]):
            naf(f'decoders.{lvl}.{i}.', dchan)
    for i in range(cfg['middle_blk_num']):
        naf(f'middle_blks.{i}.', chan)
    uchan = chan
    for lvl in range(len(cfg['dec_blk_nums'])):
        S[f'ups.{lvl}.0.weight'] = (2 * uchan, uchan, 1, 1)
        uchan //= 2
    c = width
    for lvl in range(n_enc):
        S[f'downs.{lvl}.weight'] = (2 * c, c, 2, 2); S[f'downs.{lvl}.bias'] = (2 * c,)
        c *= 2
    return S


def synth_params(cfg, seed=0, gate_std=0.1):
    """Deterministic synthetic weights that do NOT depend on nn.Module init
    order: tensor i (registration order) ~ U(-b,b), b=1/sqrt(fan_in) for convs
    (PyTorch-default-like scale), LN weight 1+0.1n / bias 0.1n, beta/gamma
    N(0,gate_std) so blocks are not identities."""
    P = OrderedDict()
    for i, (name, shape) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        if name.endswith('beta') or name.endswith('gamma'):
            t = torch.randn(shape, generator=g) * gate_std
        elif 'norm' in name:
            t = torch.randn(shape, generator=g) * 0.1
            if name.endswith('weight'):
                t = t + 1.0
        elif name.endswith('weight'):
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * b
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        P[name] = t
    return P


def synth_pair(B, H, W, seed=1234, sigma=15.0, ref_hw=None):
    """SURVEY 8d synthetic inputs: gt = clamp(bicubic-up(U[0,1] at 1/32 res)),
    ref = gt, lq = gt + N(0,(sigma/255)^2)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(H // 32, 2), max(W // 32, 2), generator=g)
    gt = F.interpolate(low, size=(H, W), mode='bicubic', align_corners=False).clamp(0, 1)
    lq = gt + torch.randn(B, 3, H, W, generator=g) * (sigma / 255.0)
    if ref_hw is None:
        ref = gt.clone()
    else:
        low_r = torch.rand(B, 3, max(ref_hw[0] // 32, 2), max(ref_hw[1] // 32, 2), generator=g)
        ref = F.interpolate(low_r, size=ref_hw, mode='bicubic', align_corners=False).clamp(0, 1)
    return lq, gt, ref


# --------------------------------------------------------------------------
# a24 L1 loss (losses/losses.py:11-13,26-53), a26 PSNR (metrics/psnr_ssim.py:9-63,
# utils/utils_image.py:129-192), a25 scheduler (models/lr_scheduler.py:186-232)
# --------------------------------------------------------------------------

def l1_loss(pred, gt, loss_weight=1.0):
    return loss_weight * (pred - gt).abs().mean()


def tensor_to_uint8_img(t):
    t = t.detach().float().clamp(0, 1)
    return np.round(t.numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)


def psnr(img1, img2, crop_border=0):
    a = np.asarray(img1, dtype=np.float64)
    b = np.asarray(img2, dtype=np.float64)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    peak = 1.0 if a.max() <= 1 else 255.0
    return 20.0 * np.log10(peak / np.sqrt(mse))


def cosine_restart_cyclic_lr(t, base_lr, periods, restart_weights, eta_mins):
    """LR at scheduler epoch t (last_epoch) for one param group."""
    cum = np.cumsum(periods)
    idx = next(i for i, p in enumerate(cum) if t <= p)
    start = 0 if idx == 0 else cum[idx - 1]
    return eta_mins[idx] + restart_weights[idx] * 0.5 * (base_lr - eta_mins[idx]) * \
        (1 + math.cos(math.pi * ((t - start) / periods[idx])))


# --------------------------------------------------------------------------
# a19/a21/a23  train step: clip_grad_norm_(0.01) + AdamW with two LR groups
#   models/image_restoration_ref_model.py:141-181, 199-284
# --------------------------------------------------------------------------

class OracleTrainer:
    """State = params + AdamW moments; step() = forward, L1, backward,
    global-norm clip (max_norm), decoupled-weight-decay Adam update with the
    'masa' substring LR split.  AdamW restated explicitly (no torch.optim)."""

    def __init__(self, P, cfg, lr=2e-4, ref_lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999),
                 eps=1e-8, max_norm=0.01, use_grad_clip=True, forward_fn=None):
        self.P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
        self.cfg = cfg
        self.forward_fn = forward_fn or nafnet_ref_forward      # e.g. restormer_ref_oracle.restormer_ref_forward
        self.lr = {k: (ref_lr if 'masa' in k else lr) for k in self.P}
        self.base_lr, self.base_ref_lr = lr, ref_lr
        self.wd, self.betas, self.eps = weight_decay, betas, eps
        self.max_norm, self.use_grad_clip = max_norm, use_grad_clip
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.t = 0

    def set_lrs(self, lr, ref_lr):
        self.lr = {k: (ref_lr if 'masa' in k else lr) for k in self.P}

    def step(self, lq, gt, ref):
        for p in self.P.values():
            p.grad = None
        out = self.forward_fn(self.P, self.cfg, lq, ref)
        loss = l1_loss(out, gt)
        loss.backward()
        # a tensor the network never uses keeps grad None: clip_grad_norm_ and torch.optim.AdamW (decay included) skip it
        grads = {k: p.grad for k, p in self.P.items() if p.grad is not None}
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        if self.use_grad_clip:
            coef = min(1.0, float(self.max_norm / (total + 1e-6)))
        else:
            coef = 1.0
        self.t += 1
        b1, b2 = self.betas
        with torch.no_grad():
            for k, p in self.P.items():
                if k not in grads:
                    continue
                g = grads[k] * coef
                lr = self.lr[k]
                p.mul_(1 - lr * self.wd)
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1 = 1 - b1 ** self.t
                bc2 = 1 - b2 ** self.t
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-lr / bc1)
        return float(loss.detach()), float(total), out.detach()

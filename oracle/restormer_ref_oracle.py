"""CPU ORACLE (test infrastructure, NOT product code).

Functional torch-CPU fp32 restatement of the reference's Restormer-ref guided-restoration network
(SURVEY.md 8a rows a13-a18): `models/archs/network_restormer_guided_arch.py`.  Pure functions over a
parameter dict keyed by the reference's state-dict names; shares no code with the reference.  The MASA
front-end (encoder, searches, transfer) is the one restated in oracle/nafnet_ref_oracle.py, used here
with the 4-level pyramid [L1..L4] and padder_size 8 (:546).

Reference defect R1: `RestormerRefFusion.forward` indexes the encoder pyramid as
feat[1..4] = 1/1 .. 1/8 scale, but this file's 4-level `Encoder.forward` (:99-133) returns [L1..L4] at
feat[0..3]; as written the reference raises an IndexError.  The only assignment under which the code
runs is feat[k] = L_k (k = 1..4): the golden generator wraps Encoder.forward to return
[None, L1, L2, L3, L4], and this restatement uses [L1..L4] directly.

Pinned against the reference itself: tests/golden/restormer_*.npz are produced by
tests/golden/make_golden_restormer.py (imports the reference in the build container);
tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import nafnet_ref_oracle as NO

LN_EPS = 1e-5


# --------------------------------------------------------------------------
# a13  BiasFree_LayerNorm / WithBias_LayerNorm (:172-218)
# --------------------------------------------------------------------------
def layernorm(x, P, pre, ln_type):
    """x [B,C,H,W]; statistics over C per pixel, biased variance, eps 1e-5.  BiasFree divides the
    UNcentred x by the standard deviation (:188-190)."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    w = P[pre + 'body.weight'].view(1, -1, 1, 1)
    if ln_type == 'BiasFree':
        return x / torch.sqrt(var + LN_EPS) * w
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + P[pre + 'body.bias'].view(1, -1, 1, 1)


# --------------------------------------------------------------------------
# a14  FeedForward / GDFN (:223-241)
# --------------------------------------------------------------------------
def gdfn(x, P, pre):
    t = F.conv2d(x, P[pre + 'project_in.weight'], P.get(pre + 'project_in.bias'))
    t = F.conv2d(t, P[pre + 'dwconv.weight'], P.get(pre + 'dwconv.bias'), padding=1, groups=t.shape[1])
    x1, x2 = t.chunk(2, dim=1)
    return F.conv2d(F.gelu(x1) * x2, P[pre + 'project_out.weight'], P.get(pre + 'project_out.bias'))


# --------------------------------------------------------------------------
# a15  Attention / MDTA (:246-277)
# --------------------------------------------------------------------------
def mdta(x, P, pre, heads):
    b, c, h, w = x.shape
    t = F.conv2d(x, P[pre + 'qkv.weight'], P.get(pre + 'qkv.bias'))
    t = F.conv2d(t, P[pre + 'qkv_dwconv.weight'], P.get(pre + 'qkv_dwconv.bias'), padding=1, groups=3 * c)
    q, k, v = t.chunk(3, dim=1)
    q = q.reshape(b, heads, c // heads, h * w)
    k = k.reshape(b, heads, c // heads, h * w)
    v = v.reshape(b, heads, c // heads, h * w)
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = (q @ k.transpose(-2, -1)) * P[pre + 'temperature']
    attn = attn.softmax(dim=-1)
    out = (attn @ v).reshape(b, c, h, w)
    return F.conv2d(out, P[pre + 'project_out.weight'], P.get(pre + 'project_out.bias'))


# --------------------------------------------------------------------------
# a16  TransformerBlock (:318-331), TransformerResFusionBlock (:334-353)
# --------------------------------------------------------------------------
def transformer_block(x, P, pre, heads, ln_type):
    x = x + mdta(layernorm(x, P, pre + 'norm1.', ln_type), P, pre + 'attn.', heads)
    x = x + gdfn(layernorm(x, P, pre + 'norm2.', ln_type), P, pre + 'ffn.')
    return x


def fusion_block(x, P, pre, heads, ln_type):
    return transformer_block(x, P, pre, heads, ln_type) * P[pre + 'alpha'] + x


def block_sequence(x, P, pre, n, heads, ln_type, fusion=False):
    for i in range(n):
        x = (fusion_block if fusion else transformer_block)(x, P, f'{pre}{i}.', heads, ln_type)
    return x


# --------------------------------------------------------------------------
# a17  OverlapPatchEmbed / Downsample / Upsample (:358-391)
# --------------------------------------------------------------------------
def downsample(x, P, pre):
    return F.pixel_unshuffle(F.conv2d(x, P[pre + 'body.0.weight'], padding=1), 2)


def upsample(x, P, pre):
    return F.pixel_shuffle(F.conv2d(x, P[pre + 'body.0.weight'], padding=1), 2)


# --------------------------------------------------------------------------
# a18  RestormerRefFusion.forward (:751-963)
# --------------------------------------------------------------------------
def default_cfg(**kw):
    cfg = dict(inp_channels=3, out_channels=3, dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
               heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias',
               nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1], lr_block_size=8,
               ref_down_block_size=1.5, dilations=[1, 2, 3], psize=3)
    cfg.update(kw)
    return cfg


PADDER = 8          # self.padder_size = 2 ** 3 (:546)


def restormer_ref_forward(P, cfg, inp, ref, return_aux=False):
    H0, W0 = inp.shape[-2:]
    mult = PADDER * cfg['lr_block_size']
    inp = NO.pad_to_multiple(inp, mult)
    ref = NO.pad_to_multiple(ref, mult)
    h, w = inp.shape[-2:]
    hr, wr = ref.shape[-2:]
    feat_lq = NO.masa_encoder(inp, P, 'masa_enc.', cfg['ext_n_blocks'], levels=4)
    feat_ref = NO.masa_encoder(ref, P, 'masa_enc.', cfg['ext_n_blocks'], levels=4)
    res = NO.masa_match_and_transfer(feat_lq, feat_ref, cfg, h, w, hr, wr, return_aux, padder=PADDER)
    warp, aux = res if return_aux else (res, None)
    hd, ln, nb, nfz = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg['reffusion_n_blocks']
    dim = cfg['dim']

    x = F.conv2d(inp, P['patch_embed.proj.weight'], P.get('patch_embed.proj.bias'), padding=1)
    x = block_sequence(torch.cat([x, warp[0]], 1), P, 'masa_blk_enc_level1.', nfz[0], hd[0], ln, True)[:, :dim]
    e1 = block_sequence(x, P, 'encoder_level1.', nb[0], hd[0], ln)
    x = downsample(e1, P, 'down1_2.')
    x = block_sequence(torch.cat([x, warp[1]], 1), P, 'masa_blk_enc_level2.', nfz[1], hd[1], ln, True)[:, :2 * dim]
    e2 = block_sequence(x, P, 'encoder_level2.', nb[1], hd[1], ln)
    x = downsample(e2, P, 'down2_3.')
    x = block_sequence(torch.cat([x, warp[2]], 1), P, 'masa_blk_enc_level3.', nfz[2], hd[2], ln, True)[:, :4 * dim]
    e3 = block_sequence(x, P, 'encoder_level3.', nb[2], hd[2], ln)
    x = downsample(e3, P, 'down3_4.')
    x = block_sequence(torch.cat([x, warp[3]], 1), P, 'masa_blk_enc_level4.', nfz[3], hd[3], ln, True)[:, :8 * dim]
    x = block_sequence(x, P, 'latent.', nb[3], hd[3], ln)

    x = torch.cat([upsample(x, P, 'up4_3.'), e3], 1)
    x = F.conv2d(x, P['reduce_chan_level3.weight'], P.get('reduce_chan_level3.bias'))
    x = block_sequence(x, P, 'decoder_level3.', nb[2], hd[2], ln)
    x = torch.cat([upsample(x, P, 'up3_2.'), e2], 1)
    x = F.conv2d(x, P['reduce_chan_level2.weight'], P.get('reduce_chan_level2.bias'))
    x = block_sequence(x, P, 'decoder_level2.', nb[1], hd[1], ln)
    x = torch.cat([upsample(x, P, 'up2_1.'), e1], 1)
    x = block_sequence(x, P, 'decoder_level1.', nb[0], hd[0], ln)
    x = block_sequence(x, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln)
    x = F.conv2d(x, P['output.weight'], P.get('output.bias'), padding=1) + inp
    out = x[:, :, :H0, :W0]
    if return_aux:
        aux['warp'] = warp
        return out, aux
    return out


# --------------------------------------------------------------------------
# parameters (names / shapes / registration order of RestormerRefFusion.__init__, :504-676)
# --------------------------------------------------------------------------
def param_shapes(cfg):
    S = OrderedDict()
    nf, dim, ic = cfg['nf'], cfg['dim'], cfg['inp_channels']
    bias, ln = cfg['bias'], cfg['LayerNorm_type']
    ext = cfg['ext_n_blocks']
    cnt = [ext[0], ext[1], ext[2], ext[2]]
    cin = ic
    for k in range(1, 5):
        c = nf * 2 ** (k - 1)
        S[f'masa_enc.conv_L{k}.weight'] = (c, cin, 3, 3)
        S[f'masa_enc.conv_L{k}.bias'] = (c,)
        for i in range(cnt[k - 1]):
            for j in (1, 2):
                S[f'masa_enc.blk_L{k}.{i}.conv{j}.weight'] = (c, c, 3, 3)
                S[f'masa_enc.blk_L{k}.{i}.conv{j}.bias'] = (c,)
        cin = c

    def conv(name, co, ci, k, b=bias):
        S[name + '.weight'] = (co, ci, k, k)
        if b:
            S[name + '.bias'] = (co,)

    def norm(pre, c):
        S[pre + 'body.weight'] = (c,)
        if ln != 'BiasFree':
            S[pre + 'body.bias'] = (c,)

    def block(pre, c, heads, fusion=False):
        if fusion:
            S[pre + 'alpha'] = (1,)
        norm(pre + 'norm1.', c)
        S[pre + 'attn.temperature'] = (heads, 1, 1)
        conv(pre + 'attn.qkv', 3 * c, c, 1)
        S[pre + 'attn.qkv_dwconv.weight'] = (3 * c, 1, 3, 3)
        if bias:
            S[pre + 'attn.qkv_dwconv.bias'] = (3 * c,)
        conv(pre + 'attn.project_out', c, c, 1)
        norm(pre + 'norm2.', c)
        hid = int(c * cfg['ffn_expansion_factor'])
        conv(pre + 'ffn.project_in', 2 * hid, c, 1)
        S[pre + 'ffn.dwconv.weight'] = (2 * hid, 1, 3, 3)
        if bias:
            S[pre + 'ffn.dwconv.bias'] = (2 * hid,)
        conv(pre + 'ffn.project_out', c, hid, 1)

    def seq(pre, n, c, heads, fusion=False):
        for i in range(n):
            block(f'{pre}{i}.', c, heads, fusion)

    hd, nb, nfz = cfg['heads'], cfg['num_blocks'], cfg['reffusion_n_blocks']
    conv('patch_embed.proj', dim, ic, 3, b=False)           # OverlapPatchEmbed default bias=False (:360)
    seq('masa_blk_enc_level1.', nfz[0], 2 * dim, hd[0], True)
    seq('encoder_level1.', nb[0], dim, hd[0])
    conv('down1_2.body.0', dim // 2, dim, 3, b=False)
    seq('masa_blk_enc_level2.', nfz[1], 4 * dim, hd[1], True)
    seq('encoder_level2.', nb[1], 2 * dim, hd[1])
    conv('down2_3.body.0', dim, 2 * dim, 3, b=False)
    seq('masa_blk_enc_level3.', nfz[2], 8 * dim, hd[2], True)
    seq('encoder_level3.', nb[2], 4 * dim, hd[2])
    conv('down3_4.body.0', 2 * dim, 4 * dim, 3, b=False)
    seq('masa_blk_enc_level4.', nfz[3], 16 * dim, hd[3], True)
    seq('latent.', nb[3], 8 * dim, hd[3])
    conv('up4_3.body.0', 16 * dim, 8 * dim, 3, b=False)
    conv('reduce_chan_level3', 4 * dim, 8 * dim, 1)
    seq('decoder_level3.', nb[2], 4 * dim, hd[2])
    conv('up3_2.body.0', 8 * dim, 4 * dim, 3, b=False)
    conv('reduce_chan_level2', 2 * dim, 4 * dim, 1)
    seq('decoder_level2.', nb[1], 2 * dim, hd[1])
    conv('up2_1.body.0', 4 * dim, 2 * dim, 3, b=False)
    seq('decoder_level1.', nb[0], 2 * dim, hd[0])
    seq('refinement.', cfg['num_refinement_blocks'], 2 * dim, hd[0])
    conv('output', cfg['out_channels'], 2 * dim, 3)
    return S


def synth_params(cfg, seed=0, alpha_std=0.1):
    """deterministic synthetic weights (independent of nn.Module init order): convs U(-b,b) with
    b = 1/sqrt(fan_in), LN weight 1+0.1n / bias 0.1n, temperature 1+0.2n, alpha N(0, alpha_std)
    so the fusion blocks are not identities."""
    P = OrderedDict()
    for i, (name, shape) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed(seed * 100003 + 7919 + i)
        if name.endswith('alpha'):
            t = torch.randn(shape, generator=g) * alpha_std
        elif name.endswith('temperature'):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif 'norm' in name:
            t = torch.randn(shape, generator=g) * 0.1
            if name.endswith('weight'):
                t = t + 1.0
        elif name.endswith('weight'):
            b = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
            t = (torch.rand(shape, generator=g) * 2 - 1) * b
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        P[name] = t
    return P

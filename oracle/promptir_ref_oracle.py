"""CPU ORACLE (test infrastructure, NOT product code).

Functional torch-CPU fp32 restatement of the reference's PromptIR-ref guided-restoration network (first
of the "next" architectures): `models/archs/network_promptir_guided_arch.py`.  Pure functions over a parameter dict
keyed by the reference's state-dict names; shares no code with the reference.  Its LayerNorm / MDTA / GDFN /
TransformerBlock / TransformerResFusionBlock / Downsample / Upsample classes (:176-400) are the same as Restormer-ref's
and are taken from oracle/restormer_ref_oracle.py; the MASA front-end is the one of oracle/nafnet_ref_oracle.py with the
4-level pyramid and padder_size 8 (:631).  New here: PromptGenBlock (:417-441) and the prompt decoder wiring (:1057-1092).

Two reference defects decide what can be pinned (both verified by running the reference in the build container):
  R1  `PromptIRRefFusion.forward` indexes the 4-entry encoder pyramid at feat[1..4] (:897-898, :958-977) -> IndexError as
      written; as for Restormer-ref the only assignment under which the code runs is feat[k] = L_k.
  R4  with `decoder=False` -- the value in the reference's own YAML (001_promptir_all_in_one_restoration.yml:57) -- the
      384-channel latent is fed to `up4_3 = Upsample(dim*4)` whose conv expects 192 channels (:733, :1065): RuntimeError.
      Only `decoder=True` runs, and its hard-wired prompt widths (64/128/320, +192/+224/+512, :643-645, :734-757) fix
      dim = nf = 48.  This oracle therefore restates the decoder=True network.
`chnl_reduce1-3` and `reduce_noise_channel_1-3` (:647-651, :668, :687) are registered but never used: they are part of
the state dict and receive no gradient.

Pinned against the reference itself: tests/golden/promptir_*.npz are produced by tests/golden/make_golden_promptir.py
(imports the reference in the build container, R1 wrapped as above); tests/test_oracle_golden.py checks this file
against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import nafnet_ref_oracle as NO
from . import restormer_ref_oracle as RO

PADDER = 8          # self.padder_size = 2 ** 3 (:631)
# (prompt_dim, prompt_len, prompt_size, lin_dim) of prompt1..3 (:643-645)
PROMPTS = {1: (64, 5, 64, 96), 2: (128, 5, 32, 192), 3: (320, 5, 16, 384)}


def prompt_gen(x, P, pre):
    """PromptGenBlock.forward (:424-441): softmax-weighted sum of the prompt components, resized, 3x3 conv."""
    B, C, H, W = x.shape
    emb = x.mean(dim=(-2, -1))
    w = F.softmax(F.linear(emb, P[pre + 'linear_layer.weight'], P[pre + 'linear_layer.bias']), dim=1)     # [B, L]
    comp = P[pre + 'prompt_param'][0]                                                                       # [L, D, S, S]
    prompt = (w.view(B, -1, 1, 1, 1) * comp.unsqueeze(0)).sum(dim=1)
    prompt = F.interpolate(prompt, (H, W), mode='bilinear')
    return F.conv2d(prompt, P[pre + 'conv3x3.weight'], padding=1)


def default_cfg(**kw):
    cfg = dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
               heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', decoder=True,
               nf=48, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1], lr_block_size=8,
               ref_down_block_size=1.5, dilations=[1, 2, 3], psize=3)
    cfg.update(kw)
    return cfg


def promptir_ref_forward(P, cfg, inp, ref, return_aux=False):
    if not cfg.get('decoder', True) or cfg['dim'] != 48 or cfg['nf'] != 48:
        raise ValueError('PromptIR-ref runs only with decoder=True and dim = nf = 48 (reference defect R4, module docstring)')
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
    seq = RO.block_sequence

    def pw(x, name):
        return F.conv2d(x, P[name + '.weight'], P.get(name + '.bias'))

    x = F.conv2d(inp, P['patch_embed.proj.weight'], P.get('patch_embed.proj.bias'), padding=1)
    x = seq(torch.cat([x, warp[0]], 1), P, 'masa_blk_enc_level1.', nfz[0], hd[0], ln, True)[:, :dim]
    e1 = seq(x, P, 'encoder_level1.', nb[0], hd[0], ln)
    x = RO.downsample(e1, P, 'down1_2.')
    x = seq(torch.cat([x, warp[1]], 1), P, 'masa_blk_enc_level2.', nfz[1], hd[1], ln, True)[:, :2 * dim]
    e2 = seq(x, P, 'encoder_level2.', nb[1], hd[1], ln)
    x = RO.downsample(e2, P, 'down2_3.')
    x = seq(torch.cat([x, warp[2]], 1), P, 'masa_blk_enc_level3.', nfz[2], hd[2], ln, True)[:, :4 * dim]
    e3 = seq(x, P, 'encoder_level3.', nb[2], hd[2], ln)
    x = RO.downsample(e3, P, 'down3_4.')
    x = seq(torch.cat([x, warp[3]], 1), P, 'masa_blk_enc_level4.', nfz[3], hd[3], ln, True)[:, :8 * dim]
    x = seq(x, P, 'latent.', nb[3], hd[3], ln)

    # prompt decoder (:1057-1092); the three noise_level blocks all use heads[2] (:736, :747, :757)
    x = torch.cat([x, prompt_gen(x, P, 'prompt3.')], 1)
    x = pw(RO.transformer_block(x, P, 'noise_level3.', hd[2], ln), 'reduce_noise_level3')
    x = torch.cat([RO.upsample(x, P, 'up4_3.'), e3], 1)
    x = seq(pw(x, 'reduce_chan_level3'), P, 'decoder_level3.', nb[2], hd[2], ln)
    x = torch.cat([x, prompt_gen(x, P, 'prompt2.')], 1)
    x = pw(RO.transformer_block(x, P, 'noise_level2.', hd[2], ln), 'reduce_noise_level2')
    x = torch.cat([RO.upsample(x, P, 'up3_2.'), e2], 1)
    x = seq(pw(x, 'reduce_chan_level2'), P, 'decoder_level2.', nb[1], hd[1], ln)
    x = torch.cat([x, prompt_gen(x, P, 'prompt1.')], 1)
    x = pw(RO.transformer_block(x, P, 'noise_level1.', hd[2], ln), 'reduce_noise_level1')
    x = torch.cat([RO.upsample(x, P, 'up2_1.'), e1], 1)
    x = seq(x, P, 'decoder_level1.', nb[0], hd[0], ln)
    x = seq(x, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln)
    x = F.conv2d(x, P['output.weight'], P.get('output.bias'), padding=1) + inp
    out = x[:, :, :H0, :W0]
    if return_aux:
        aux['warp'] = warp
        return out, aux
    return out


# --------------------------------------------------------------------------
# parameters (names / shapes of PromptIRRefFusion.__init__, :594-757, decoder=True)
# --------------------------------------------------------------------------
UNUSED = ('chnl_reduce1', 'chnl_reduce2', 'chnl_reduce3', 'reduce_noise_channel_1', 'reduce_noise_channel_2',
          'reduce_noise_channel_3')


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
    conv('patch_embed.proj', dim, ic, 3, b=False)
    for k in (1, 2, 3):
        pd, pl, ps, ld = PROMPTS[k]
        S[f'prompt{k}.prompt_param'] = (1, pl, pd, ps, ps)
        S[f'prompt{k}.linear_layer.weight'] = (pl, ld)
        S[f'prompt{k}.linear_layer.bias'] = (pl,)
        S[f'prompt{k}.conv3x3.weight'] = (pd, pd, 3, 3)
    conv('chnl_reduce1', 64, 64, 1)
    conv('chnl_reduce2', 128, 128, 1)
    conv('chnl_reduce3', 256, 320, 1)
    conv('reduce_noise_channel_1', dim, dim + 64, 1)
    seq('masa_blk_enc_level1.', nfz[0], 2 * dim, hd[0], True)
    seq('encoder_level1.', nb[0], dim, hd[0])
    conv('down1_2.body.0', dim // 2, dim, 3, b=False)
    conv('reduce_noise_channel_2', 2 * dim, 2 * dim + 128, 1)
    seq('masa_blk_enc_level2.', nfz[1], 4 * dim, hd[1], True)
    seq('encoder_level2.', nb[1], 2 * dim, hd[1])
    conv('down2_3.body.0', dim, 2 * dim, 3, b=False)
    conv('reduce_noise_channel_3', 4 * dim, 4 * dim + 256, 1)
    seq('masa_blk_enc_level3.', nfz[2], 8 * dim, hd[2], True)
    seq('encoder_level3.', nb[2], 4 * dim, hd[2])
    conv('down3_4.body.0', 2 * dim, 4 * dim, 3, b=False)
    seq('masa_blk_enc_level4.', nfz[3], 16 * dim, hd[3], True)
    seq('latent.', nb[3], 8 * dim, hd[3])
    conv('up4_3.body.0', 8 * dim, 4 * dim, 3, b=False)
    conv('reduce_chan_level3', 4 * dim, 2 * dim + 192, 1)
    block('noise_level3.', 4 * dim + 512, hd[2])
    conv('reduce_noise_level3', 4 * dim, 4 * dim + 512, 1)
    seq('decoder_level3.', nb[2], 4 * dim, hd[2])
    conv('up3_2.body.0', 8 * dim, 4 * dim, 3, b=False)
    conv('reduce_chan_level2', 2 * dim, 4 * dim, 1)
    block('noise_level2.', 2 * dim + 224, hd[2])
    conv('reduce_noise_level2', 4 * dim, 2 * dim + 224, 1)
    seq('decoder_level2.', nb[1], 2 * dim, hd[1])
    conv('up2_1.body.0', 4 * dim, 2 * dim, 3, b=False)
    block('noise_level1.', 2 * dim + 64, hd[2])
    conv('reduce_noise_level1', 2 * dim, 2 * dim + 64, 1)
    seq('decoder_level1.', nb[0], 2 * dim, hd[0])
    seq('refinement.', cfg['num_refinement_blocks'], 2 * dim, hd[0])
    conv('output', cfg['out_channels'], 2 * dim, 3)
    return S


def synth_params(cfg, seed=0, alpha_std=0.1):
    """deterministic synthetic weights: convs / linears U(-b, b) with b = 1/sqrt(fan_in), LN weight 1 + 0.1 n / bias 0.1 n,
    temperature 1 + 0.2 n, alpha N(0, alpha_std) so the fusion blocks are not identities, prompt components U(0, 1) as the
    reference initialises them (:420), linear biases spread so the softmax is not uniform."""
    P = OrderedDict()
    for i, (name, shape) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed(seed * 100003 + 104729 + i)
        if name.endswith('alpha'):
            t = torch.randn(shape, generator=g) * alpha_std
        elif name.endswith('temperature'):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith('prompt_param'):
            t = torch.rand(shape, generator=g)
        elif 'linear_layer' in name:
            t = torch.randn(shape, generator=g) * (0.5 if name.endswith('bias') else 2.0 / math.sqrt(shape[-1]))
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


def loss_and_grads(P, cfg, inp, ref, gt):
    """L1 loss (mean) and the gradients of every USED parameter (autograd over this restatement)."""
    Pg = OrderedDict((k, v.clone().requires_grad_(not k.startswith(UNUSED))) for k, v in P.items())
    out = promptir_ref_forward(Pg, cfg, inp, ref)
    loss = (out - gt).abs().mean()
    names = [k for k, v in Pg.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [Pg[k] for k in names])
    return out.detach(), loss.detach(), OrderedDict(zip(names, gs))

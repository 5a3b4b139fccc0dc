"""Hand-written forward/backward of PromptIR-ref (models/archs/network_promptir_guided_arch.py of the reference) on the
HIP kernels -- SURVEY.md 8f, the first of the "next" guided architectures.

The file's LayerNorm / MDTA / GDFN / TransformerBlock / TransformerResFusionBlock / Downsample / Upsample classes
(:176-400) are the ones of Restormer-ref, so everything up to the latent is restormer_engine's code path (same MASA
front-end: 4-level pyramid, padder_size 8, reference defect R1).  New here: PromptGenBlock (:417-441, csrc/tdr_prompt.hip)
and the prompt decoder (:1057-1092).  Only the `decoder=True`, dim = nf = 48 network exists: with `decoder=False` the
reference itself raises in up4_3 (defect R4, oracle/promptir_ref_oracle.py), and the prompt widths are hard-wired.
`chnl_reduce1-3` / `reduce_noise_channel_1-3` are registered by the reference but never used: they never reach this
engine (no gradient, as in the reference where their .grad stays None).
"""
import torch

from . import engine as E
from . import kernels as K
from . import restormer_engine as R

PADDER_LOG2 = 3      # self.padder_size = 2 ** 3 (:631)


# ---------------------------------------------------------------------------
# PromptGenBlock (:417-441)
# ---------------------------------------------------------------------------
def prompt_fwd(x, P, pre):
    """x [N,C,H,W] -> prompt [N,D,H,W].  The softmax-weighted sum over the L components and the bilinear resize are both
    linear and commute: the L*D parameter planes are resized once per call, not once per image."""
    N, C, H, W = x.shape
    comp = P[pre + 'prompt_param'][0]                                   # [L, D, S, S]
    L, D, S, _ = comp.shape
    emb = K.plane_mean(x)
    w = K.prompt_weights_fwd(emb, P[pre + 'linear_layer.weight'], P[pre + 'linear_layer.bias'])
    Pi = comp if (H, W) == (S, S) else K.resize_bilinear(comp.contiguous(), H, W)
    mix = K.prompt_mix_fwd(w, Pi.contiguous())
    out = E.conv_fwd(mix, P[pre + 'conv3x3.weight'], None, 1, 1)
    return out, (x.shape, emb, w, Pi, mix)


def prompt_bwd(dout, P, pre, saved, G):
    """returns demb-broadcast information (demb [N,C], 1/HW): the caller adds it to the gradient of x."""
    (N, C, H, W), emb, w, Pi, mix = saved
    comp = P[pre + 'prompt_param']
    L, D, S = comp.shape[1], comp.shape[2], comp.shape[3]
    dmix, G[pre + 'conv3x3.weight'], _ = E.conv_bwd(dout, mix, P[pre + 'conv3x3.weight'], 1, 1, bias=False)
    dPi, dw = K.prompt_mix_bwd(w, Pi.contiguous(), dmix.contiguous())
    dcomp = dPi if (H, W) == (S, S) else K.resize_bilinear_bwd(dPi, S, S)
    G[pre + 'prompt_param'] = dcomp.view(1, L, D, S, S)
    G[pre + 'linear_layer.weight'], G[pre + 'linear_layer.bias'], demb = K.prompt_weights_bwd(
        emb, P[pre + 'linear_layer.weight'], w, dw)
    return demb, 1.0 / (H * W)


def _prompt_stage_fwd(x, P, k, heads, ln):
    """cat([x, prompt_k(x)]) -> noise_level_k (TransformerBlock) -> reduce_noise_level_k (1x1)   (:1057-1084)"""
    pr, sv_p = prompt_fwd(x, P, f'prompt{k}.')
    cat = K.concat2(x, pr)
    t, sv_t = R.tblock_fwd(cat, E._sub(P, f'noise_level{k}.'), heads, ln)
    y = R._pw_fwd(t, P, f'reduce_noise_level{k}')
    return y, (x.shape[1], sv_p, sv_t, t)


def _prompt_stage_bwd(d, P, k, heads, ln, saved, G):
    c, sv_p, sv_t, t = saved
    d = R._pw_bwd(d, t, P, f'reduce_noise_level{k}', G)
    E.set_late_prefix(f'noise_level{k}.')
    dcat, g = R.tblock_bwd(d, E._sub(P, f'noise_level{k}.'), heads, ln, sv_t)
    E.set_late_prefix('')
    E._put(G, f'noise_level{k}.', g)
    dx = K.slice_channels(dcat, 0, c)
    demb, inv = prompt_bwd(K.slice_channels(dcat, c, dcat.shape[1]), P, f'prompt{k}.', sv_p, G)
    return K.plane_add_(dx, demb, inv)


# ---------------------------------------------------------------------------
# whole network  PromptIRRefFusion.forward (:864-1092)
# ---------------------------------------------------------------------------
def net_fwd(P, cfg, inp, ref):
    """ref = None: the UN-GUIDED `PromptIR` of the same file (:443-590): no MASA pyramid, no fusion blocks, no padding (sizes must
    be multiples of 8).  Like the guided class it only exists as decoder=True, dim = 48 (with decoder=False its up4_3 receives
    the 384-channel latent on a 192-channel convolution and the reference raises: R4)."""
    guided = ref is not None
    if not cfg.get('decoder', True) or cfg['dim'] != 48 or (guided and cfg['nf'] != 48):
        raise ValueError('PromptIR(-ref) exists only as decoder=True, dim = nf = 48 (the reference raises otherwise: defect R4)')
    N = inp.shape[0]
    if guided:
        pyr, (H0, W0, Hp, Wp) = E.pyramids_fwd(P, cfg, inp, ref, PADDER_LOG2, 4)
        warp, sv_masa = E.masa_fwd(pyr.lq_deep, pyr.ref_feats, N, pyr.geo)
    else:
        H0, W0 = inp.shape[2:]
        if H0 % 8 or W0 % 8:
            raise ValueError(f'PromptIR: H, W must be multiples of 8 (three PixelUnshuffle(2) stages); got {H0}x{W0}')
        Hp, Wp = H0, W0
        import types
        pyr, warp, sv_masa = types.SimpleNamespace(inp_p=inp.contiguous(), geo=None), None, None
    inp_p, geo = pyr.inp_p, pyr.geo
    hd, ln, nb, nfz, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg.get('reffusion_n_blocks'), cfg['dim']

    x = E.conv_fwd(inp_p, P['patch_embed.proj.weight'], P.get('patch_embed.proj.bias'), 1, 1)
    sv_lv, enc_out = [], []
    for l in range(4):
        c = dim * 2 ** l
        sv_f = None
        if guided:
            f, sv_f = R.seq_fwd(K.concat2(x, warp[l]), P, R._FUS[l], nfz[l], hd[l], ln, fusion=True)
            x = K.slice_channels(f, 0, c)
        e, sv_e = R.seq_fwd(x, P, R._ENC[l], nb[l], hd[l], ln)
        enc_out.append(e)
        sv_lv.append((sv_f, sv_e))
        if l < 3:
            x = R.down_fwd(e, P[R._DOWN[l]])
    e1, e2, e3, lat = enc_out
    # ---- prompt decoder: the three noise_level blocks all use heads[2] (:736, :747, :757)
    p3, sv_p3 = _prompt_stage_fwd(lat, P, 3, hd[2], ln)
    cat3 = K.concat2(R.up_fwd(p3, P['up4_3.body.0.weight']), e3)
    d3, sv_d3 = R.seq_fwd(R._pw_fwd(cat3, P, 'reduce_chan_level3'), P, 'decoder_level3.', nb[2], hd[2], ln)
    p2, sv_p2 = _prompt_stage_fwd(d3, P, 2, hd[2], ln)
    cat2 = K.concat2(R.up_fwd(p2, P['up3_2.body.0.weight']), e2)
    d2, sv_d2 = R.seq_fwd(R._pw_fwd(cat2, P, 'reduce_chan_level2'), P, 'decoder_level2.', nb[1], hd[1], ln)
    p1, sv_p1 = _prompt_stage_fwd(d2, P, 1, hd[2], ln)
    cat1 = K.concat2(R.up_fwd(p1, P['up2_1.body.0.weight']), e1)
    d1, sv_d1 = R.seq_fwd(cat1, P, 'decoder_level1.', nb[0], hd[0], ln)
    rf, sv_rf = R.seq_fwd(d1, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln)
    out_p = E.conv_fwd(rf, P['output.weight'], P.get('output.bias'), 1, 1, res=inp_p)
    out = out_p if (Hp, Wp) == (H0, W0) else K.pad_crop(out_p, H0, W0)
    saved = (N, (H0, W0, Hp, Wp), geo, pyr, None, None, sv_masa, sv_lv, enc_out, sv_p3, p3, cat3, sv_d3, sv_p2, p2, cat2,
             sv_d2, sv_p1, p1, sv_d1, rf, sv_rf)
    return out, saved


def net_bwd(dout, P, cfg, saved, G=None):
    G = {} if G is None else G
    with E.deferred_join(), E.late_leaves(G):       # (leaf 1x1 weight gradients: engine.DEFER_WGRAD)
        return _net_bwd(dout, P, cfg, saved, G)


def _net_bwd(dout, P, cfg, saved, G):
    (N, (H0, W0, Hp, Wp), geo, pyr, _, _, sv_masa, sv_lv, enc_out, sv_p3, p3, cat3, sv_d3, sv_p2, p2, cat2, sv_d2,
     sv_p1, p1, sv_d1, rf, sv_rf) = saved
    G = {} if G is None else G
    hd, ln, nb, nfz, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg.get('reffusion_n_blocks'), cfg['dim']
    inp_p = pyr.inp_p
    dout = dout.contiguous()
    if (Hp, Wp) != (H0, W0):
        dout = K.pad_crop(dout, Hp, Wp)
    has_ob = 'output.bias' in P
    d, G['output.weight'], db = E.conv_bwd(dout, rf, P['output.weight'], 1, 1, bias=has_ob)
    if has_ob:
        G['output.bias'] = db
    d = R.seq_bwd(d, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln, sv_rf, G)
    d = R.seq_bwd(d, P, 'decoder_level1.', nb[0], hd[0], ln, sv_d1, G)          # grad of cat[up(p1), e1]
    de1 = d[:, dim:]
    d, G['up2_1.body.0.weight'] = R.up_bwd(K.slice_channels(d, 0, dim), p1, P['up2_1.body.0.weight'])
    d = _prompt_stage_bwd(d, P, 1, hd[2], ln, sv_p1, G)
    d = R.seq_bwd(d, P, 'decoder_level2.', nb[1], hd[1], ln, sv_d2, G)
    d = R._pw_bwd(d, cat2, P, 'reduce_chan_level2', G)
    de2 = d[:, 2 * dim:]
    d, G['up3_2.body.0.weight'] = R.up_bwd(K.slice_channels(d, 0, 2 * dim), p2, P['up3_2.body.0.weight'])
    d = _prompt_stage_bwd(d, P, 2, hd[2], ln, sv_p2, G)
    d = R.seq_bwd(d, P, 'decoder_level3.', nb[2], hd[2], ln, sv_d3, G)
    d = R._pw_bwd(d, cat3, P, 'reduce_chan_level3', G)
    de3 = d[:, 2 * dim:]                                                        # cat3 = [up(p3): 2*dim | e3: 4*dim]
    d, G['up4_3.body.0.weight'] = R.up_bwd(K.slice_channels(d, 0, 2 * dim), p3, P['up4_3.body.0.weight'])
    d = _prompt_stage_bwd(d, P, 3, hd[2], ln, sv_p3, G)
    dskip = [de1, de2, de3]
    dwarp = [None] * 4
    for l in reversed(range(4)):
        c = dim * 2 ** l
        sv_f, sv_e = sv_lv[l]
        d = R.seq_bwd(d, P, R._ENC[l], nb[l], hd[l], ln, sv_e, G)
        dx = d
        if sv_f is not None:
            df = torch.zeros(N, 2 * c, d.shape[2], d.shape[3], dtype=torch.float32, device=d.device)
            K.copy_rows(d, c * d.shape[2] * d.shape[3], df, 2 * c * d.shape[2] * d.shape[3], N, c * d.shape[2] * d.shape[3])
            dcat = R.seq_bwd(df, P, R._FUS[l], nfz[l], hd[l], ln, sv_f, G, fusion=True)
            dwarp[l] = dcat[:, c:]
            dx = K.slice_channels(dcat, 0, c)
        if l > 0:
            d, G[R._DOWN[l - 1]] = R.down_bwd(dx, enc_out[l - 1], P[R._DOWN[l - 1]])
            d = K.add_(d, dskip[l - 1])
        else:
            has_pb = 'patch_embed.proj.bias' in P
            _, G['patch_embed.proj.weight'], db = E.conv_bwd(dx, inp_p, P['patch_embed.proj.weight'], 1, 1, need_dx=False,
                                                             bias=has_pb)
            if has_pb:
                G['patch_embed.proj.bias'] = db
    E.run_late_leaves(G, (lambda: E.pyramids_bwd(dwarp, pyr, P, cfg, sv_masa, G)) if sv_masa is not None else (lambda: None))
    return G

"""Hand-written forward/backward of Restormer-ref (models/archs/network_restormer_guided_arch.py of the
reference) on the HIP kernels -- SURVEY.md 8a rows a13-a18.

Same conventions as engine.py (whose MASA front-end, dense-conv helpers and encoder this reuses): `*_fwd`
returns (out, saved), `*_bwd` returns (dx, grads); parameters travel as dicts keyed by the reference's
state-dict names.  No ATen arithmetic runs on the device: every tensor op is a call into libtdr_hip.so.

Reference defect R1: `RestormerRefFusion.forward` indexes the 4-level encoder pyramid one slot off (feat[4]
of a 4-entry list, :790-793,832-846).  The only assignment under which that code runs is feat[k] = L_k;
this engine uses [L1..L4] with padder_size 8 (:546), exactly what the golden generator's wrapped Encoder does.

MDTA (:246-277) on this hardware: the "tokens" are channels, so both contractions over the H*W pixels are
convolution-shaped -- q k^T is a per-image weight-gradient GEMM (tdr_conv_wgrad, per_image), attn v is a 1x1
convolution with per-image weights -- and run on the MFMA kernels; csrc/tdr_mdta.hip does the c x c softmax
algebra and emits the weights already in the packed layout those kernels read.
"""
import torch

from . import engine as E
from . import kernels as K
from .kernels import EPI_PSHUF, PACK_DGRAD_S1, PACK_FWD

LN_EPS = 1e-5        # :190,208
PADDER_LOG2 = 3      # self.padder_size = 2 ** 3 (:546)


# ---------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------
def _ln_fwd(x, P, pre, ln_type):
    """LayerNorm (:211-218): BiasFree (:172-190) scales the uncentred x; WithBias (:193-208)."""
    center = ln_type != 'BiasFree'
    return K.layernorm2d_fwd(x, P[pre + 'body.weight'], P[pre + 'body.bias'] if center else None, LN_EPS, center=center)


def _ln_bwd(go, x, mu, rs, P, pre, ln_type, G, add=None):
    center = ln_type != 'BiasFree'
    gx, gw, gb = K.layernorm2d_bwd(go, x, mu, rs, P[pre + 'body.weight'], add=add, center=center)
    G[pre + 'body.weight'] = gw
    if center:
        G[pre + 'body.bias'] = gb
    return gx


def _pw_fwd(x, P, name, res=None):
    """1x1 conv `name` (bias optional)."""
    w = P[name + '.weight']
    wp, mp, *_ = K.pack_weights(w, PACK_FWD)
    return K.conv_forward(x, wp, mp, w.shape[0], 1, bias=P.get(name + '.bias'), res=res)


def _pw_bwd(dout, x, P, name, G):
    """returns dx; stores dW (and db)."""
    w = P[name + '.weight']
    Cout, Cin = w.shape[0], w.shape[1]
    has_b = (name + '.bias') in P
    # parameter gradient: a leaf off the data-gradient chain (engine._leaf_wgrad1x1: deferred, leaves of one shape in one grouped launch)
    def post(g, db):
        return {name + '.weight': g.view(Cout, Cin, 1, 1), name + '.bias': db} if has_b else {name + '.weight': g.view(Cout, Cin, 1, 1)}
    E._leaf_wgrad1x1((x, dout), (x, dout, Cout, Cin, False), post, G, want_db=has_b)
    wp, mp, *_ = K.pack_weights(w, PACK_DGRAD_S1)
    return K.conv_forward(dout, wp, mp, Cin, 1)


def _img_conv(x, Wt, Cout, out=None):
    """1x1 conv with per-image weights Wt [N, Kp, Mp] (fp32 packed layout emitted by tdr_mdta_*): on the split-bf16
    kernel after one batched re-pack, or directly on the exact fp32 kernel (kernels.MATH)."""
    Kp, Mp = Wt.shape[-2], Wt.shape[-1]
    if K.MATH != 'f32':
        pw, per_b = K.pack_f32packed_to_bx3(Wt)
        return K.conv_forward(x, pw, Mp, Cout, 1, wp_ns=per_b, out=out)
    return K.conv_forward(x, Wt, Mp, Cout, 1, wp_ns=Kp * Mp, out=out)


# ---------------------------------------------------------------------------
# TransformerBlock (:318-331) = x + MDTA(LN(x)); + GDFN(LN(.))
# ---------------------------------------------------------------------------
def tblock_fwd(x, P, heads, ln_type):
    N, Cc, H, W = x.shape
    xn, mu1, rs1 = _ln_fwd(x, P, 'norm1.', ln_type)
    # ---- MDTA (:246-277)
    t = _pw_fwd(xn, P, 'attn.qkv')                                               # [N,3C,H,W]
    qkv = K.dwconv_fwd(t, P['attn.qkv_dwconv.weight'], P.get('attn.qkv_dwconv.bias'))
    ss = K.row_sumsq(qkv, 2 * Cc)                                                # |q_i|^2, |k_j|^2
    Gm = K.conv_wgrad(qkv[:, Cc:2 * Cc], qkv[:, :Cc], Cc, Cc, 1, per_image=True, fp16_range=True).view(N, Cc, Cc)   # q k^T
    A, AT = K.mdta_softmax(Gm, ss, P['attn.temperature'], heads)
    o = _img_conv(qkv[:, 2 * Cc:], AT, Cc)                                       # attn v
    y = _pw_fwd(o, P, 'attn.project_out', res=x)
    # ---- GDFN (:223-241)
    yn, mu2, rs2 = _ln_fwd(y, P, 'norm2.', ln_type)
    t2 = _pw_fwd(yn, P, 'ffn.project_in')                                        # [N,2h,H,W]
    g = K.dwgelu_fwd(t2, P['ffn.dwconv.weight'], P.get('ffn.dwconv.bias'))
    out = _pw_fwd(g, P, 'ffn.project_out', res=y)
    return out, (x, xn, mu1, rs1, t, qkv, ss, Gm, A, o, y, yn, mu2, rs2, t2, g)


def tblock_bwd(dout, P, heads, ln_type, saved):
    x, xn, mu1, rs1, t, qkv, ss, Gm, A, o, y, yn, mu2, rs2, t2, g = saved
    N, Cc, H, W = x.shape
    G = {}
    # ---- GDFN
    dg = _pw_bwd(dout, g, P, 'ffn.project_out', G)
    b = P.get('ffn.dwconv.bias')
    dt2, G['ffn.dwconv.weight'], db = K.dwgelu_bwd(dg, t2, P['ffn.dwconv.weight'], b)
    if b is not None:
        G['ffn.dwconv.bias'] = db
    dyn = _pw_bwd(dt2, yn, P, 'ffn.project_in', G)
    dy = _ln_bwd(dyn, y, mu2, rs2, P, 'norm2.', ln_type, G, add=dout)
    # ---- MDTA
    do = _pw_bwd(dy, o, P, 'attn.project_out', G)
    dA = K.conv_wgrad(qkv[:, 2 * Cc:], do, Cc, Cc, 1, per_image=True).view(N, Cc, Cc)      # dA_ij = do_i . v_j
    Wm, G['attn.temperature'] = K.mdta_bwd(Gm, ss, P['attn.temperature'], A, dA, heads)
    dqkv = torch.empty_like(qkv)
    _img_conv(do, A, Cc, out=dqkv[:, 2 * Cc:])                                             # dv = attn^T do
    _img_conv(qkv[:, :2 * Cc], Wm, 2 * Cc, out=dqkv[:, :2 * Cc])                           # d[q;k] = W [q;k]
    has_b = 'attn.qkv_dwconv.bias' in P
    dt, G['attn.qkv_dwconv.weight'], db = K.dwconv_bwd(dqkv, t, P['attn.qkv_dwconv.weight'], want_db=has_b)
    if has_b:
        G['attn.qkv_dwconv.bias'] = db
    dxn = _pw_bwd(dt, xn, P, 'attn.qkv', G)
    dx = _ln_bwd(dxn, x, mu1, rs1, P, 'norm1.', ln_type, G, add=dy)
    E.maybe_join()
    return dx, G


# TransformerResFusionBlock (:334-353): block(x) * alpha + x
def fblock_fwd(x, P, heads, ln_type):
    z, sv = tblock_fwd(x, P, heads, ln_type)
    return K.axpby_dev(z, P['alpha'], x), (sv, z)


def fblock_bwd(dout, P, heads, ln_type, saved):
    sv, z = saved
    dalpha = K.dot(dout, z)
    dz = K.axpby_dev(dout, P['alpha'])
    with E.deferred_join():
        dx, G = tblock_bwd(dz, P, heads, ln_type, sv)
    G['alpha'] = dalpha
    dx = K.add_(dx, dout)
    E.maybe_join()
    return dx, G


def seq_fwd(x, P, pre, n, heads, ln_type, fusion=False):
    saved = []
    for i in range(n):
        x, sv = (fblock_fwd if fusion else tblock_fwd)(x, E._sub(P, f'{pre}{i}.'), heads, ln_type)
        saved.append(sv)
    return x, saved


def seq_bwd(d, P, pre, n, heads, ln_type, saved, G, fusion=False):
    for i in reversed(range(n)):
        E.set_late_prefix(f'{pre}{i}.')
        d, g = (fblock_bwd if fusion else tblock_bwd)(d, E._sub(P, f'{pre}{i}.'), heads, ln_type, saved[i])
        E._put(G, f'{pre}{i}.', g)
    E.set_late_prefix('')              # (top-level leaves -- reduce_chan_level*, skip_conv -- carry full names)
    return d


# ---------------------------------------------------------------------------
# Downsample / Upsample (:370-391): 3x3 conv (no bias) + PixelUnshuffle(2) / PixelShuffle(2)
# ---------------------------------------------------------------------------
def down_fwd(x, w):
    return K.pixel_unshuffle2(E.conv_fwd(x, w, None, 1, 1))


def down_bwd(dout, x, w):
    dx, dw, _ = E.conv_bwd(K.pixel_shuffle2(dout), x, w, 1, 1, bias=False)
    return dx, dw


def up_fwd(x, w):
    wp, mp, *_ = K.pack_weights(w, PACK_FWD)
    return K.conv_forward(x, wp, mp, w.shape[0], 3, pad=1, epi=EPI_PSHUF)


def up_bwd(dout, x, w):
    dx, dw, _ = E.conv_bwd(K.pixel_unshuffle2(dout), x, w, 1, 1, bias=False)
    return dx, dw


# ---------------------------------------------------------------------------
# whole network  RestormerRefFusion.forward (:751-963)
# ---------------------------------------------------------------------------
_FUS = ['masa_blk_enc_level1.', 'masa_blk_enc_level2.', 'masa_blk_enc_level3.', 'masa_blk_enc_level4.']
_ENC = ['encoder_level1.', 'encoder_level2.', 'encoder_level3.', 'latent.']
_DOWN = ['down1_2.body.0.weight', 'down2_3.body.0.weight', 'down3_4.body.0.weight']


def net_fwd(P, cfg, inp, ref):
    """inp, ref [N,3,H,W] -> (out [N,3,H,W], saved).  cfg: constructor kwargs of RestormerRefFusion."""
    N = inp.shape[0]
    pyr, (H0, W0, Hp, Wp) = E.pyramids_fwd(P, cfg, inp, ref, PADDER_LOG2, 4)
    inp_p, geo = pyr.inp_p, pyr.geo
    warp, sv_masa = E.masa_fwd(pyr.lq_deep, pyr.ref_feats, N, geo)
    hd, ln, nb, nfz, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg['reffusion_n_blocks'], cfg['dim']

    x = E.conv_fwd(inp_p, P['patch_embed.proj.weight'], P.get('patch_embed.proj.bias'), 1, 1)
    sv_lv, enc_out = [], []
    for l in range(4):
        c = dim * 2 ** l
        f, sv_f = seq_fwd(K.concat2(x, warp[l]), P, _FUS[l], nfz[l], hd[l], ln, fusion=True)
        x = K.slice_channels(f, 0, c)                      # `[:, :embed_dim // 2]` (:892,903,914,925)
        if l == 0:
            x_l1 = x                                       # `inp_enc_level1`: what skip_conv reads when dual_pixel_task (:958)
        e, sv_e = seq_fwd(x, P, _ENC[l], nb[l], hd[l], ln)
        enc_out.append(e)
        sv_lv.append((sv_f, sv_e))
        if l < 3:
            x = down_fwd(e, P[_DOWN[l]])
    e1, e2, e3, lat = enc_out
    cat3 = K.concat2(up_fwd(lat, P['up4_3.body.0.weight']), e3)
    d3, sv_d3 = seq_fwd(_pw_fwd(cat3, P, 'reduce_chan_level3'), P, 'decoder_level3.', nb[2], hd[2], ln)
    cat2 = K.concat2(up_fwd(d3, P['up3_2.body.0.weight']), e2)
    d2, sv_d2 = seq_fwd(_pw_fwd(cat2, P, 'reduce_chan_level2'), P, 'decoder_level2.', nb[1], hd[1], ln)
    cat1 = K.concat2(up_fwd(d2, P['up2_1.body.0.weight']), e1)
    d1, sv_d1 = seq_fwd(cat1, P, 'decoder_level1.', nb[0], hd[0], ln)
    rf, sv_rf = seq_fwd(d1, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln)
    if cfg.get('dual_pixel_task'):
        # dual-pixel defocus deblurring (:955-959): output(refined + skip_conv(inp_enc_level1)), no `+ inp_img`
        rf = _pw_fwd(x_l1, P, 'skip_conv', res=rf)
        out_p = E.conv_fwd(rf, P['output.weight'], P.get('output.bias'), 1, 1)
    else:
        out_p = E.conv_fwd(rf, P['output.weight'], P.get('output.bias'), 1, 1, res=inp_p)
    out = out_p if (Hp, Wp) == (H0, W0) else K.pad_crop(out_p, H0, W0)
    saved = (N, (H0, W0, Hp, Wp), geo, pyr, x_l1, None, sv_masa, sv_lv, enc_out, cat3, d3, sv_d3, cat2, d2, sv_d2,
             sv_d1, rf, sv_rf)
    return out, saved


def net_bwd(dout, P, cfg, saved, G=None):
    G = {} if G is None else G
    with E.deferred_join(), E.late_leaves(G):
        return _net_bwd(dout, P, cfg, saved, G)


def _net_bwd(dout, P, cfg, saved, G):
    (N, (H0, W0, Hp, Wp), geo, pyr, x_l1, _, sv_masa, sv_lv, enc_out, cat3, d3, sv_d3, cat2, d2, sv_d2, sv_d1, rf,
     sv_rf) = saved
    hd, ln, nb, nfz, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg['reffusion_n_blocks'], cfg['dim']
    e1, e2, e3, lat = enc_out
    inp_p = pyr.inp_p
    dout = dout.contiguous()
    if (Hp, Wp) != (H0, W0):
        dout = K.pad_crop(dout, Hp, Wp)
    has_ob = 'output.bias' in P
    d, G['output.weight'], db = E.conv_bwd(dout, rf, P['output.weight'], 1, 1, bias=has_ob)
    if has_ob:
        G['output.bias'] = db
    dskip_l1 = _pw_bwd(d, x_l1, P, 'skip_conv', G) if cfg.get('dual_pixel_task') else None
    d = seq_bwd(d, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln, sv_rf, G)
    d = seq_bwd(d, P, 'decoder_level1.', nb[0], hd[0], ln, sv_d1, G)            # grad of cat[up(d2), e1]
    de1 = d[:, dim:]
    d, G['up2_1.body.0.weight'] = up_bwd(K.slice_channels(d, 0, dim), d2, P['up2_1.body.0.weight'])
    d = seq_bwd(d, P, 'decoder_level2.', nb[1], hd[1], ln, sv_d2, G)
    d = _pw_bwd(d, cat2, P, 'reduce_chan_level2', G)
    de2 = d[:, 2 * dim:]
    d, G['up3_2.body.0.weight'] = up_bwd(K.slice_channels(d, 0, 2 * dim), d3, P['up3_2.body.0.weight'])
    d = seq_bwd(d, P, 'decoder_level3.', nb[2], hd[2], ln, sv_d3, G)
    d = _pw_bwd(d, cat3, P, 'reduce_chan_level3', G)
    de3 = d[:, 4 * dim:]
    d, G['up4_3.body.0.weight'] = up_bwd(K.slice_channels(d, 0, 4 * dim), lat, P['up4_3.body.0.weight'])
    dskip = [de1, de2, de3]
    dwarp = [None] * 4
    for l in reversed(range(4)):
        c = dim * 2 ** l
        sv_f, sv_e = sv_lv[l]
        d = seq_bwd(d, P, _ENC[l], nb[l], hd[l], ln, sv_e, G)
        if l == 0 and dskip_l1 is not None:
            d = K.add_(d, dskip_l1)
        df = torch.zeros(N, 2 * c, d.shape[2], d.shape[3], dtype=torch.float32, device=d.device)
        K.copy_rows(d, c * d.shape[2] * d.shape[3], df, 2 * c * d.shape[2] * d.shape[3], N, c * d.shape[2] * d.shape[3])
        dcat = seq_bwd(df, P, _FUS[l], nfz[l], hd[l], ln, sv_f, G, fusion=True)
        dwarp[l] = dcat[:, c:]
        dx = K.slice_channels(dcat, 0, c)
        if l > 0:
            d, G[_DOWN[l - 1]] = down_bwd(dx, enc_out[l - 1], P[_DOWN[l - 1]])
            d = K.add_(d, dskip[l - 1])
        else:
            has_pb = 'patch_embed.proj.bias' in P
            _, G['patch_embed.proj.weight'], db = E.conv_bwd(dx, inp_p, P['patch_embed.proj.weight'], 1, 1, need_dx=False,
                                                             bias=has_pb)
            if has_pb:
                G['patch_embed.proj.bias'] = db
    E.run_late_leaves(G, lambda: E.pyramids_bwd(dwarp, pyr, P, cfg, sv_masa, G))
    return G


# ---------------------------------------------------------------------------
# un-guided Restormer.forward (:464-501): the same blocks without the reference branch
# ---------------------------------------------------------------------------
def unet_fwd(P, cfg, inp):
    """inp [N, inp_channels, H, W], H and W multiples of 8 (the reference has no padding here: its PixelUnshuffle raises
    otherwise) -> (out, saved)"""
    N, _, H, W = inp.shape
    if H % 8 or W % 8:
        raise ValueError(f'Restormer: H, W must be multiples of 8 (three PixelUnshuffle(2) stages, :370-378); got {H}x{W}')
    hd, ln, nb, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg['dim']
    inp = inp.contiguous()
    x0 = E.conv_fwd(inp, P['patch_embed.proj.weight'], P.get('patch_embed.proj.bias'), 1, 1)
    x, sv_lv, enc_out = x0, [], []
    for l in range(4):
        e, sv_e = seq_fwd(x, P, _ENC[l], nb[l], hd[l], ln)
        enc_out.append(e)
        sv_lv.append(sv_e)
        if l < 3:
            x = down_fwd(e, P[_DOWN[l]])
    e1, e2, e3, lat = enc_out
    cat3 = K.concat2(up_fwd(lat, P['up4_3.body.0.weight']), e3)
    d3, sv_d3 = seq_fwd(_pw_fwd(cat3, P, 'reduce_chan_level3'), P, 'decoder_level3.', nb[2], hd[2], ln)
    cat2 = K.concat2(up_fwd(d3, P['up3_2.body.0.weight']), e2)
    d2, sv_d2 = seq_fwd(_pw_fwd(cat2, P, 'reduce_chan_level2'), P, 'decoder_level2.', nb[1], hd[1], ln)
    cat1 = K.concat2(up_fwd(d2, P['up2_1.body.0.weight']), e1)
    d1, sv_d1 = seq_fwd(cat1, P, 'decoder_level1.', nb[0], hd[0], ln)
    rf, sv_rf = seq_fwd(d1, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln)
    if cfg.get('dual_pixel_task'):
        rf = _pw_fwd(x0, P, 'skip_conv', res=rf)
        out = E.conv_fwd(rf, P['output.weight'], P.get('output.bias'), 1, 1)
    else:
        out = E.conv_fwd(rf, P['output.weight'], P.get('output.bias'), 1, 1, res=inp)
    return out, (inp, x0, sv_lv, enc_out, cat3, d3, sv_d3, cat2, d2, sv_d2, sv_d1, rf, sv_rf)


def unet_bwd(dout, P, cfg, saved, G=None):
    """-> G (parameter gradients; the input image is data)"""
    with E.deferred_join():
        inp, x0, sv_lv, enc_out, cat3, d3, sv_d3, cat2, d2, sv_d2, sv_d1, rf, sv_rf = saved
        G = {} if G is None else G
        hd, ln, nb, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg['dim']
        e1, e2, e3, lat = enc_out
        dout = dout.contiguous()
        has_ob = 'output.bias' in P
        d, G['output.weight'], db = E.conv_bwd(dout, rf, P['output.weight'], 1, 1, bias=has_ob)
        if has_ob:
            G['output.bias'] = db
        dskip0 = _pw_bwd(d, x0, P, 'skip_conv', G) if cfg.get('dual_pixel_task') else None
        d = seq_bwd(d, P, 'refinement.', cfg['num_refinement_blocks'], hd[0], ln, sv_rf, G)
        d = seq_bwd(d, P, 'decoder_level1.', nb[0], hd[0], ln, sv_d1, G)
        de1 = d[:, dim:]
        d, G['up2_1.body.0.weight'] = up_bwd(K.slice_channels(d, 0, dim), d2, P['up2_1.body.0.weight'])
        d = seq_bwd(d, P, 'decoder_level2.', nb[1], hd[1], ln, sv_d2, G)
        d = _pw_bwd(d, cat2, P, 'reduce_chan_level2', G)
        de2 = d[:, 2 * dim:]
        d, G['up3_2.body.0.weight'] = up_bwd(K.slice_channels(d, 0, 2 * dim), d3, P['up3_2.body.0.weight'])
        d = seq_bwd(d, P, 'decoder_level3.', nb[2], hd[2], ln, sv_d3, G)
        d = _pw_bwd(d, cat3, P, 'reduce_chan_level3', G)
        de3 = d[:, 4 * dim:]
        d, G['up4_3.body.0.weight'] = up_bwd(K.slice_channels(d, 0, 4 * dim), lat, P['up4_3.body.0.weight'])
        dskip = [de1, de2, de3]
        for l in reversed(range(4)):
            d = seq_bwd(d, P, _ENC[l], nb[l], hd[l], ln, sv_lv[l], G)
            if l > 0:
                d, G[_DOWN[l - 1]] = down_bwd(d, enc_out[l - 1], P[_DOWN[l - 1]])
                d = K.add_(d, dskip[l - 1])
            else:
                if dskip0 is not None:
                    d = K.add_(d, dskip0)
                has_pb = 'patch_embed.proj.bias' in P
                _, G['patch_embed.proj.weight'], db = E.conv_bwd(d, inp, P['patch_embed.proj.weight'], 1, 1,
                                                                 need_dx=False, bias=has_pb)
                if has_pb:
                    G['patch_embed.proj.bias'] = db
        return G

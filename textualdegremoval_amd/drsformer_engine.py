"""Hand-written forward/backward of DRSformer-ref without the MEFC sub-network (`DRSformer200L_SPA_RefFusion`,
models/archs/network_drsformer_guided_arch_200L_SPA.py of the reference) on the HIP kernels -- SURVEY.md 8f, second "next"
architecture.  Topology = Restormer-ref minus the refinement stage (restormer_engine's MASA front-end, dense convs,
Down/Upsample, LayerNorm are reused); the blocks differ:

* Top-K Sparse Attention (:257-328): the four masked softmaxes all multiply the same v, so they collapse into ONE c x c
  matrix A = sum_m attn_m * softmax(topk_m(logits)) (tdr_tksa_softmax) and the MDTA data flow is unchanged: q k^T as a
  per-image weight-gradient GEMM, A v and d[q;k] = W [q;k] as per-image 1x1 convs (tdr_tksa_bwd emits W, dtemperature, dattn_m).
* mixed-scale feed-forward (:213-253): 3x3 and 5x5 depthwise branches with ReLU, cross-concatenated, then the grouped
  2->1 convs (Conv2d(2h, h, groups=h)) with ReLU: tdr_dwk_fwd / tdr_dwk_bwd.

Reference defects (oracle/drsformer_ref_oracle.py): R1 (pyramid index), R5 (missing import, construction fails as shipped),
R6: the level-1 reference fusion is computed and discarded -- here it is not computed at all; `masa_blk_enc_level1.*` never
reach the engine (no gradient, as in the reference).
"""
import os

import torch

from . import engine as E
from . import kernels as K
from . import restormer_engine as R

PADDER_LOG2 = 3


def _am(P):
    """attn1..attn4 as one [4] device vector"""
    am = torch.empty(4, dtype=torch.float32, device=P['attn.attn1'].device)
    for m in range(4):
        K.copy_rows(P[f'attn.attn{m + 1}'], 0, am[m:m + 1], 0, 1, 1)
    return am


def attn_fwd(xn, P, heads, res=None):
    """Attention.forward (:274-328) on the normalised input; `res` is added to the projection (the block's skip)."""
    N, Cc, H, W = xn.shape
    t = R._pw_fwd(xn, P, 'attn.qkv')
    qkv = K.dwconv_fwd(t, P['attn.qkv_dwconv.weight'], P.get('attn.qkv_dwconv.bias'))
    ss = K.row_sumsq(qkv, 2 * Cc)
    Gm = K.conv_wgrad(qkv[:, Cc:2 * Cc], qkv[:, :Cc], Cc, Cc, 1, per_image=True, fp16_range=True).view(N, Cc, Cc)
    am = _am(P)
    A, AT = K.tksa_softmax(Gm, ss, P['attn.temperature'], am, heads)
    o = R._img_conv(qkv[:, 2 * Cc:], AT, Cc)
    y = R._pw_fwd(o, P, 'attn.project_out', res=res)
    return y, (xn, t, qkv, ss, Gm, am, A, o)


def attn_bwd(dy, P, heads, saved, G):
    xn, t, qkv, ss, Gm, am, A, o = saved
    N, Cc, H, W = xn.shape
    do = R._pw_bwd(dy, o, P, 'attn.project_out', G)
    dA = K.conv_wgrad(qkv[:, 2 * Cc:], do, Cc, Cc, 1, per_image=True).view(N, Cc, Cc)
    Wm, G['attn.temperature'], dam = K.tksa_bwd(Gm, ss, P['attn.temperature'], am, dA, heads)
    for m in range(4):
        G[f'attn.attn{m + 1}'] = dam[m:m + 1]
    dqkv = torch.empty_like(qkv)
    R._img_conv(do, A, Cc, out=dqkv[:, 2 * Cc:])
    R._img_conv(qkv[:, :2 * Cc], Wm, 2 * Cc, out=dqkv[:, :2 * Cc])
    has_b = 'attn.qkv_dwconv.bias' in P
    dt, G['attn.qkv_dwconv.weight'], db = K.dwconv_bwd(dqkv, t, P['attn.qkv_dwconv.weight'], want_db=has_b)
    if has_b:
        G['attn.qkv_dwconv.bias'] = db
    return R._pw_bwd(dt, xn, P, 'attn.qkv', G)


def _split_ok(t2, h):
    """the in-place cross-concatenation needs 16-byte aligned plane slices and the 3x3 stencil route (W % 4 == 0)"""
    return t2.shape[-1] % 4 == 0 and (h * t2.shape[2] * t2.shape[3]) % 4 == 0 and not K.DWK_GENERIC and \
        os.environ.get('TDR_DWSG_TWO_PASS', '0') != '1'


def ffn_fwd(yn, P, res=None):
    """FeedForward.forward (:240-253)."""
    t2 = R._pw_fwd(yn, P, 'ffn.project_in')                                         # [N, 2h, H, W]
    h = t2.shape[1] // 2
    # x1 = [a3[:h] | a5[:h]], x2 = [a3[h:] | a5[h:]] (:244-247) are written in place: the 3x3 stencil owns the plane pair (c, c + h)
    # and stores its two planes to x1 / x2; the 5x5 conv runs per half.  a3 / a5 only exist as these slices.
    x1, x2 = torch.empty_like(t2), torch.empty_like(t2)
    w5, b5 = P['ffn.dwconv5x5.weight'], P.get('ffn.dwconv5x5.bias')
    if _split_ok(t2, h):
        a3 = K.dwk_fwd(t2, P['ffn.dwconv3x3.weight'], P.get('ffn.dwconv3x3.bias'), relu=True, out=(x1[:, :h], x2[:, :h]))
        K.dwk_fwd(t2[:, :h], w5[:h], None if b5 is None else b5[:h], relu=True, out=x1[:, h:])
        K.dwk_fwd(t2[:, h:], w5[h:], None if b5 is None else b5[h:], relu=True, out=x2[:, h:])
        a5 = (x1[:, h:], x2[:, h:])
    else:
        a3 = K.dwk_fwd(t2, P['ffn.dwconv3x3.weight'], P.get('ffn.dwconv3x3.bias'), relu=True)
        a5 = K.dwk_fwd(t2, w5, b5, relu=True)
        x1 = K.concat2(a3[:, :h], a5[:, :h])
        x2 = K.concat2(a3[:, h:], a5[:, h:])
    cat = torch.empty(t2.shape[0], 2 * h, t2.shape[2], t2.shape[3], dtype=torch.float32, device=t2.device)
    z1 = K.dwk_fwd(x1, P['ffn.dwconv3x3_1.weight'], P.get('ffn.dwconv3x3_1.bias'), relu=True, out=cat[:, :h])   # straight into
    z2 = K.dwk_fwd(x2, P['ffn.dwconv5x5_1.weight'], P.get('ffn.dwconv5x5_1.bias'), relu=True, out=cat[:, h:])   # the concatenation
    out = R._pw_fwd(cat, P, 'ffn.project_out', res=res)
    return out, (yn, t2, a3, a5, x1, x2, z1, z2, cat)


def ffn_bwd(dout, P, saved, G):
    yn, t2, a3, a5, x1, x2, z1, z2, cat = saved
    h = t2.shape[1] // 2

    def put(name, dw, db):
        G[name + '.weight'] = dw
        if db is not None:
            G[name + '.bias'] = db
    dcat = R._pw_bwd(dout, cat, P, 'ffn.project_out', G)
    dx1, dw, db = K.dwk_bwd(dcat[:, :h], z1, x1, P['ffn.dwconv3x3_1.weight'], want_db='ffn.dwconv3x3_1.bias' in P)   # batch-strided
    put('ffn.dwconv3x3_1', dw, db)                                                                                      # views, no copies
    dx2, dw, db = K.dwk_bwd(dcat[:, h:], z2, x2, P['ffn.dwconv5x5_1.weight'], want_db='ffn.dwconv5x5_1.bias' in P)
    put('ffn.dwconv5x5_1', dw, db)
    w5 = P['ffn.dwconv5x5.weight']
    if isinstance(a3, tuple):
        # da3 = [dx1[:h] | dx2[:h]], da5 = [dx1[h:] | dx2[h:]] are read where they are (split halves / channel slices)
        dt2, dw, db = K.dwk_bwd((dx1[:, :h], dx2[:, :h]), a3, t2, P['ffn.dwconv3x3.weight'], want_db='ffn.dwconv3x3.bias' in P)
        put('ffn.dwconv3x3', dw, db)
        has_b = 'ffn.dwconv5x5.bias' in P
        dw5 = torch.empty_like(w5)
        db5 = torch.empty(2 * h, dtype=torch.float32, device=t2.device) if has_b else None
        # the 5x5 branch adds its input gradient onto the 3x3 branch's inside the kernel
        K.dwk_bwd(dx1[:, h:], a5[0], t2[:, :h], w5[:h], want_db=has_b, dx_out=dt2[:, :h], dw_out=dw5[:h],
                  db_out=None if db5 is None else db5[:h], accumulate=True)
        K.dwk_bwd(dx2[:, h:], a5[1], t2[:, h:], w5[h:], want_db=has_b, dx_out=dt2[:, h:], dw_out=dw5[h:],
                  db_out=None if db5 is None else db5[h:], accumulate=True)
        put('ffn.dwconv5x5', dw5, db5)
        return R._pw_bwd(dt2, yn, P, 'ffn.project_in', G)
    else:
        da3 = K.concat2(dx1[:, :h], dx2[:, :h])
        da5 = K.concat2(dx1[:, h:], dx2[:, h:])
        dt2, dw, db = K.dwk_bwd(da3, a3, t2, P['ffn.dwconv3x3.weight'], want_db='ffn.dwconv3x3.bias' in P)
        put('ffn.dwconv3x3', dw, db)
        dt2b, dw, db = K.dwk_bwd(da5, a5, t2, w5, want_db='ffn.dwconv5x5.bias' in P)
        put('ffn.dwconv5x5', dw, db)
    return R._pw_bwd(K.add_(dt2, dt2b), yn, P, 'ffn.project_in', G)


def tblock_fwd(x, P, heads, ln_type):
    xn, mu1, rs1 = R._ln_fwd(x, P, 'norm1.', ln_type)
    y, sv_a = attn_fwd(xn, P, heads, res=x)
    yn, mu2, rs2 = R._ln_fwd(y, P, 'norm2.', ln_type)
    out, sv_f = ffn_fwd(yn, P, res=y)
    return out, (x, mu1, rs1, sv_a, y, mu2, rs2, sv_f)


def tblock_bwd(dout, P, heads, ln_type, saved):
    x, mu1, rs1, sv_a, y, mu2, rs2, sv_f = saved
    G = {}
    dyn = ffn_bwd(dout, P, sv_f, G)
    dy = R._ln_bwd(dyn, y, mu2, rs2, P, 'norm2.', ln_type, G, add=dout)
    dxn = attn_bwd(dy, P, heads, sv_a, G)
    dx = R._ln_bwd(dxn, x, mu1, rs1, P, 'norm1.', ln_type, G, add=dy)
    E.maybe_join()
    return dx, G


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
    E.set_late_prefix('')
    return d


def net_fwd(P, cfg, inp, ref):
    """ref = None: the UN-GUIDED `DRSformer` of the same file (network_drsformer_guided_arch.py:586-676): no MASA pyramid, no
    fusion blocks, no padding (its PixelUnshuffle raises on sizes that are not multiples of 8), MEFC sub-networks always."""
    N = inp.shape[0]
    guided = ref is not None
    if guided:
        pyr, (H0, W0, Hp, Wp) = E.pyramids_fwd(P, cfg, inp, ref, PADDER_LOG2, 4)
        warp, sv_masa = E.masa_fwd(pyr.lq_deep, pyr.ref_feats, N, pyr.geo)
    else:
        H0, W0 = inp.shape[2:]
        if H0 % 8 or W0 % 8:
            raise ValueError(f'DRSformer: H, W must be multiples of 8 (three PixelUnshuffle(2) stages); got {H0}x{W0}')
        Hp, Wp = H0, W0
        import types
        pyr, warp, sv_masa = types.SimpleNamespace(inp_p=inp.contiguous(), geo=None), None, None
    inp_p, geo = pyr.inp_p, pyr.geo
    hd, ln, nb, nfz, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg.get('reffusion_n_blocks'), cfg['dim']
    full = bool(cfg.get('mefc'))       # DRSformerRefFusion: MEFC sub-networks + a working level-1 fusion; else the 200L_SPA class
    x = E.conv_fwd(inp_p, P['patch_embed.proj.weight'], P.get('patch_embed.proj.bias'), 1, 1)
    x_embed, sv_m0 = x, None
    if full:
        x, sv_m0 = mefc_fwd(x, P, 'encoder_level0.')
    sv_lv, enc_out = [], []
    for l in range(4):
        c = dim * 2 ** l
        sv_f = None
        if guided and (l > 0 or full):   # R6 (200L_SPA only): the reference discards the level-1 fusion; it is not computed there
            f, sv_f = seq_fwd(K.concat2(x, warp[l]), P, R._FUS[l], nfz[l], hd[l], ln, fusion=True)
            x = K.slice_channels(f, 0, c)
        e, sv_e = seq_fwd(x, P, R._ENC[l], nb[l], hd[l], ln)
        enc_out.append(e)
        sv_lv.append((sv_f, sv_e))
        if l < 3:
            x = R.down_fwd(e, P[R._DOWN[l]])
    e1, e2, e3, lat = enc_out
    cat3 = K.concat2(R.up_fwd(lat, P['up4_3.body.0.weight']), e3)
    d3, sv_d3 = seq_fwd(R._pw_fwd(cat3, P, 'reduce_chan_level3'), P, 'decoder_level3.', nb[2], hd[2], ln)
    cat2 = K.concat2(R.up_fwd(d3, P['up3_2.body.0.weight']), e2)
    d2, sv_d2 = seq_fwd(R._pw_fwd(cat2, P, 'reduce_chan_level2'), P, 'decoder_level2.', nb[1], hd[1], ln)
    cat1 = K.concat2(R.up_fwd(d2, P['up2_1.body.0.weight']), e1)
    d1, sv_d1 = seq_fwd(cat1, P, 'decoder_level1.', nb[0], hd[0], ln)
    sv_m1 = None
    if full:
        d1, sv_m1 = mefc_fwd(d1, P, 'refinement.')
    out_p = E.conv_fwd(d1, P['output.weight'], P.get('output.bias'), 1, 1, res=inp_p)
    out = out_p if (Hp, Wp) == (H0, W0) else K.pad_crop(out_p, H0, W0)
    saved = (N, (H0, W0, Hp, Wp), geo, pyr, sv_m0, sv_m1, sv_masa, sv_lv, enc_out, cat3, d3, sv_d3, cat2, d2, sv_d2, sv_d1, d1)
    return out, saved


def net_bwd(dout, P, cfg, saved, G=None):
    G = {} if G is None else G
    with E.deferred_join(), E.late_leaves(G):       # (leaf 1x1 weight gradients: engine.DEFER_WGRAD)
        return _net_bwd(dout, P, cfg, saved, G)


def _net_bwd(dout, P, cfg, saved, G):
    (N, (H0, W0, Hp, Wp), geo, pyr, sv_m0, sv_m1, sv_masa, sv_lv, enc_out, cat3, d3, sv_d3, cat2, d2, sv_d2, sv_d1, d1) = saved
    full = bool(cfg.get('mefc'))
    G = {} if G is None else G
    hd, ln, nb, nfz, dim = cfg['heads'], cfg['LayerNorm_type'], cfg['num_blocks'], cfg.get('reffusion_n_blocks'), cfg['dim']
    e1, e2, e3, lat = enc_out
    inp_p = pyr.inp_p
    dout = dout.contiguous()
    if (Hp, Wp) != (H0, W0):
        dout = K.pad_crop(dout, Hp, Wp)
    has_ob = 'output.bias' in P
    d, G['output.weight'], db = E.conv_bwd(dout, d1, P['output.weight'], 1, 1, bias=has_ob)
    if has_ob:
        G['output.bias'] = db
    if full:
        d = mefc_bwd(d, P, 'refinement.', sv_m1, G)
    d = seq_bwd(d, P, 'decoder_level1.', nb[0], hd[0], ln, sv_d1, G)
    de1 = d[:, dim:]
    d, G['up2_1.body.0.weight'] = R.up_bwd(K.slice_channels(d, 0, dim), d2, P['up2_1.body.0.weight'])
    d = seq_bwd(d, P, 'decoder_level2.', nb[1], hd[1], ln, sv_d2, G)
    d = R._pw_bwd(d, cat2, P, 'reduce_chan_level2', G)
    de2 = d[:, 2 * dim:]
    d, G['up3_2.body.0.weight'] = R.up_bwd(K.slice_channels(d, 0, 2 * dim), d3, P['up3_2.body.0.weight'])
    d = seq_bwd(d, P, 'decoder_level3.', nb[2], hd[2], ln, sv_d3, G)
    d = R._pw_bwd(d, cat3, P, 'reduce_chan_level3', G)
    de3 = d[:, 4 * dim:]
    d, G['up4_3.body.0.weight'] = R.up_bwd(K.slice_channels(d, 0, 4 * dim), lat, P['up4_3.body.0.weight'])
    dskip = [de1, de2, de3]
    dwarp = [None] * 4
    for l in reversed(range(4)):
        c = dim * 2 ** l
        sv_f, sv_e = sv_lv[l]
        d = seq_bwd(d, P, R._ENC[l], nb[l], hd[l], ln, sv_e, G)
        if sv_f is not None:
            df = torch.zeros(N, 2 * c, d.shape[2], d.shape[3], dtype=torch.float32, device=d.device)
            K.copy_rows(d, c * d.shape[2] * d.shape[3], df, 2 * c * d.shape[2] * d.shape[3], N, c * d.shape[2] * d.shape[3])
            dcat = seq_bwd(df, P, R._FUS[l], nfz[l], hd[l], ln, sv_f, G, fusion=True)
            dwarp[l] = dcat[:, c:]
            d = K.slice_channels(dcat, 0, c)
        if l > 0:
            d, G[R._DOWN[l - 1]] = R.down_bwd(d, enc_out[l - 1], P[R._DOWN[l - 1]])
            d = K.add_(d, dskip[l - 1])
        else:
            if full:
                d = mefc_bwd(d, P, 'encoder_level0.', sv_m0, G)
            elif sv_masa is not None:
                dwarp[0] = torch.zeros(N, c, d.shape[2], d.shape[3], dtype=torch.float32, device=d.device)     # R6: unused warp level
            has_pb = 'patch_embed.proj.bias' in P
            _, G['patch_embed.proj.weight'], db = E.conv_bwd(d, inp_p, P['patch_embed.proj.weight'], 1, 1, need_dx=False, bias=has_pb)
            if has_pb:
                G['patch_embed.proj.bias'] = db
    E.run_late_leaves(G, (lambda: E.pyramids_bwd(dwarp, pyr, P, cfg, sv_masa, G)) if sv_masa is not None else (lambda: None))
    return G


# ---------------------------------------------------------------------------
# MEFC sub-network (`subnet`, network_drsformer_guided_arch.py:522-548) and the full DRSformerRefFusion (:679-1123)
# ---------------------------------------------------------------------------
OPS = ('sep_conv_1x1', 'sep_conv_3x3', 'sep_conv_5x5', 'sep_conv_7x7', 'dil_conv_3x3', 'dil_conv_5x5', 'dil_conv_7x7', 'avg_pool_3x3')
STEPS = 4


def _pw(x, w, relu=False, out=None):
    wp, mp, *_ = K.pack_weights(w, R.PACK_FWD)
    return K.conv_forward(x, wp, mp, w.shape[0], 1, relu=relu, out=out)


def _op_fwd(x, P, pre, j):
    name = OPS[j]
    if name == 'avg_pool_3x3':
        return K.avgpool3(x), None
    if name.startswith('sep_conv'):
        a = K.dwk_fwd(x, P[pre + 'op.0.weight'])
        r = _pw(a, P[pre + 'op.1.weight'], relu=True)
        c = K.dwk_fwd(r, P[pre + 'op.3.weight'])
        return _pw(c, P[pre + 'op.4.weight']), (a, r, c)
    a = K.dwk_fwd(x, P[pre + 'op.0.weight'], dil=2)
    return _pw(a, P[pre + 'op.1.weight']), (a,)


def _op_bwd(do, x, P, pre, j, sv, G):
    name = OPS[j]
    if name == 'avg_pool_3x3':
        return K.avgpool3(do.contiguous(), adjoint=True)
    if name.startswith('sep_conv'):
        a, r, c = sv
        dc = R._pw_bwd(do, c, P, pre + 'op.4', G)
        dr, G[pre + 'op.3.weight'], _ = K.dwk_bwd(dc, None, r, P[pre + 'op.3.weight'])
        da = R._pw_bwd(K.relu_bwd(dr, r), a, P, pre + 'op.1', G)
        dx, G[pre + 'op.0.weight'], _ = K.dwk_bwd(da, None, x, P[pre + 'op.0.weight'])
        return dx
    a, = sv
    da = R._pw_bwd(do, a, P, pre + 'op.1', G)
    dx, G[pre + 'op.0.weight'], _ = K.dwk_bwd(da, None, x, P[pre + 'op.0.weight'], dil=2)
    return dx


def mefc_fwd(x, P, pre):
    """x [N,C,H,W] -> subnet(x).  Gating: mean_hw -> Linear -> ReLU -> Linear -> softmax over the 8 operations of each step;
    the per-image operation weights scale the (saved, unscaled) operation outputs while they are copied into the concat."""
    N, C, H, W = x.shape
    nops = len(OPS)
    emb = K.plane_mean(x)
    h1 = K.linear_small_fwd(emb, P[pre + 'layers.0.ca_fc.0.weight'], P[pre + 'layers.0.ca_fc.0.bias'], relu=True)
    lg = K.linear_small_fwd(h1, P[pre + 'layers.0.ca_fc.2.weight'], P[pre + 'layers.0.ca_fc.2.bias'])
    wts = K.softmax_rows(lg.view(N * STEPS, nops))                      # [(n, step), op]
    wflat = wts.view(-1)
    g = pre + 'layers.1.'
    s0 = _pw(x, P[g + 'preprocess.op.0.weight'], relu=True)
    steps = []
    for i in range(STEPS):
        cat = torch.empty(N, nops * C, H, W, dtype=torch.float32, device=x.device)
        ops = []
        for j in range(nops):
            o, sv = _op_fwd(s0, P, f'{g}_ops.{i}._ops.{j}.', j)
            K.scale_copy(o, wflat[i * nops + j:], STEPS * nops, cat[:, j * C:(j + 1) * C])
            ops.append((o, sv))
        t = _pw(cat, P[f'{g}_ops.{i}._out.0.weight'], relu=True)
        s1 = K.add_relu(t, s0)
        steps.append((s0, cat, t, s1, ops))
        s0 = s1
    return s0, (x, emb, h1, wts, steps)


def mefc_bwd(dout, P, pre, saved, G):
    x, emb, h1, wts, steps = saved
    N, C, H, W = x.shape
    nops = len(OPS)
    g = pre + 'layers.1.'
    wflat = wts.view(-1)
    dwts = torch.empty_like(wts)
    dwflat = dwts.view(-1)
    d = dout.contiguous()
    for i in reversed(range(STEPS)):
        s0, cat, t, s1, ops = steps[i]
        d = K.relu_bwd(d, s1)                                           # relu(t + res)
        dt = K.relu_bwd(d, t)                                           # _out's ReLU
        dcat = R._pw_bwd(dt, cat, P, f'{g}_ops.{i}._out.0', G)
        ds0 = d                                                         # the `res` branch
        for j in range(nops):
            o, sv = ops[j]
            dsl = dcat[:, j * C:(j + 1) * C]
            K.rows_dot(dsl, o, dwflat[i * nops + j:], STEPS * nops)
            do = K.scale_copy(dsl, wflat[i * nops + j:], STEPS * nops, torch.empty(N, C, H, W, dtype=torch.float32, device=x.device))
            ds0 = K.add_(_op_bwd(do, s0, P, f'{g}_ops.{i}._ops.{j}.', j, sv, G), ds0)
        d = ds0
    dpre = K.relu_bwd(d, steps[0][0])
    dx = R._pw_bwd(dpre, x, P, g + 'preprocess.op.0', G)
    dlg = K.softmax_rows(wts, dy=dwts).view(N, STEPS * nops)
    dh1, G[pre + 'layers.0.ca_fc.2.weight'], G[pre + 'layers.0.ca_fc.2.bias'] = K.linear_small_bwd(dlg, None, h1, P[pre + 'layers.0.ca_fc.2.weight'])
    demb, G[pre + 'layers.0.ca_fc.0.weight'], G[pre + 'layers.0.ca_fc.0.bias'] = K.linear_small_bwd(dh1, h1, emb, P[pre + 'layers.0.ca_fc.0.weight'])
    E.maybe_join()
    return K.plane_add_(dx, demb, 1.0 / (H * W))

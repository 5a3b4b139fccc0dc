"""Forward / backward of the reference's UN-GUIDED `SFNet` (models/archs/network_sfnet_guided_arch.py:320-407) on the HIP kernels,
training-mode semantics (global average pools, BatchNorm2d on batch statistics with its running buffers moved in place).

Functional like engine.py: parameters and buffers in a dict keyed by the reference's state-dict names, `net_fwd` returns
([out 1/4, out 1/2, out full], saved), `net_bwd` the gradients keyed the same way.  Every tensor operation on the device is a call
into libtdr_hip.so: the convolutions on the library's MFMA kernels (engine.conv_fwd / conv_bwd: 3x3 stride 1 / 2, 1x1; the
4x4 / stride-2 transposed convolutions as 3x3 convolutions to 4 Cout channels + PixelShuffle, csrc/tdr_sfnet.hip), everything around
them on csrc/tdr_sfnet.hip (GELU, InstanceNorm2d, the Gap / Patch_ap band re-weighting, the dynamic low-pass filter with SFconv).
The guided class of the same file cannot run in the reference (defect R8); this is the network of that file that does."""
import torch

from . import engine as E
from . import kernels as K
from .kernels import EPI_PSHUF, PACK_FWD

BASE = 32
GROUPS = 8


# ------------------------------------------------------------------------------------------------- BasicConv (sfnet_arch_utils.py:76-98)
def conv_fwd(x, P, pre, k, stride=1, act=True, res=None):
    """conv (k x k, padding k // 2) + bias [+ exact GELU] -> (y, saved)"""
    w, b = P[pre + 'main.0.weight'], P[pre + 'main.0.bias']
    z = E.conv_fwd(x, w, b, stride, k // 2, res=None if act else res)
    if not act:
        return z, (x, None, k, stride)
    _, y = K.gelu_fwd(z)
    return y, (x, z, k, stride)


def conv_bwd(dy, P, pre, saved, G, need_dx=True, add_to_dx=None):
    x, z, k, stride = saved
    dz = dy if z is None else K.gelu_bwd(dy.contiguous(), z)
    dx, dw, db = E.conv_bwd(dz.contiguous(), x, P[pre + 'main.0.weight'], stride, k // 2, need_dx=need_dx, add_to_dx=add_to_dx)
    G[pre + 'main.0.weight'], G[pre + 'main.0.bias'] = dw, db
    return dx


def convt_fwd(x, P, pre):
    """ConvTranspose2d(4, stride 2, padding 1) + bias + GELU (BasicConv(transpose=True), :87)"""
    w, b = P[pre + 'main.0.weight'], P[pre + 'main.0.bias']
    Cout = w.shape[1]
    w3, _ = K.convt4_weight_to_3x3(w, None)
    wp, mp, *_ = K.pack_weights(w3, PACK_FWD)
    t = K.conv_forward(x, wp, mp, 4 * Cout, 3, pad=1, epi=EPI_PSHUF)
    z, y = K.gelu_fwd(t, bias=b)
    return y, (x, z, w3)


def convt_bwd(dy, P, pre, saved, G):
    x, z, w3 = saved
    Cin, Cout = P[pre + 'main.0.weight'].shape[:2]
    dz = K.gelu_bwd(dy.contiguous(), z)
    G[pre + 'main.0.bias'] = K.channel_sum(dz)
    dT = K.pixel_unshuffle2(dz)
    dx, dw3, _ = E.conv_bwd(dT, x, w3, 1, 1, bias=False)
    G[pre + 'main.0.weight'] = K.convt4_grad_from_3x3(dw3.contiguous(), None, Cin, Cout)[0]
    return dx


# ------------------------------------------------------------------------------------------------- dynamic_filter + SFconv (:152-236)
_training = True        # set by net_fwd for the pass it runs
_tlsc = None            # mode[0] == 'test': the TLSC base size (246 Indoor / 210 Outdoor); forward only
TLSC_BASE = {'Indoor': 246, 'Outdoor': 210}       # sfnet_arch_utils.py:110-113


def _box(x, H, W):
    """AvgPool2d(base_size) of the reference's test mode on a contiguous [N, C, H, W]: k = size * base // 256 (:33-34), replicate-padded back"""
    return K.local_avgpool(x, min(H, H * _tlsc // 256), min(W, W * _tlsc // 256))


def dyn_fwd(x, P, pre, k, out):
    """x, out: channel-slice views [N, c, H, W]"""
    ap = K.plane_mean(x)
    taps, ah, al, sv = K.sf_dyn_vec_fwd(ap, P, pre, k, GROUPS, training=_training)
    low, mix = K.sf_dynfilt_fwd(x, taps, ah, al, k, GROUPS)
    if _tlsc is not None:
        # SFconv.gap is the box mean: per-pixel attention (dynamic_filter.ap above stays global, :171); 1x1 convolutions on the map
        H, W = x.shape[2:]
        z = E.conv_fwd(_box(K.sf_emerge(x, low), H, W), P[pre + 'modulate.fc.weight'], P[pre + 'modulate.fc.bias'], 1, 0)
        lh = E.conv_fwd(z, P[pre + 'modulate.fcs.0.weight'], P[pre + 'modulate.fcs.0.bias'], 1, 0)
        ll = E.conv_fwd(z, P[pre + 'modulate.fcs.1.weight'], P[pre + 'modulate.fcs.1.bias'], 1, 0)
        mix = K.sf_softmax_mix(x, low, lh, ll)
    E.conv_fwd(mix, P[pre + 'modulate.out.weight'], P[pre + 'modulate.out.bias'], 1, 0, out=out)
    return (x, low, mix, taps, ah, al, sv)


def dyn_bwd(dy, P, pre, k, saved, dx, G):
    x, low, mix, taps, ah, al, sv = saved
    dmix, dwo, dbo = E.conv_bwd(dy.contiguous(), mix, P[pre + 'modulate.out.weight'], 1, 0)
    G[pre + 'modulate.out.weight'], G[pre + 'modulate.out.bias'] = dwo, dbo
    dah, dal, dtaps = K.sf_dynfilt_bwd_reduce(dmix, x, low, ah, al, k, GROUPS)
    dap, gv = K.sf_dyn_vec_bwd(dtaps, dah, dal, P, pre, k, sv, GROUPS)
    for n, g in gv.items():
        G[pre + n] = g
    K.sf_dynfilt_bwd_dx(dmix, taps, ah, al, dap, k, dx, GROUPS)


# ------------------------------------------------------------------------------------------------- ResBlock (:120-149)
def res_fwd(x, P, pre, filt):
    N, C, H, W = x.shape
    h = C // 2
    y1, sv1 = conv_fwd(x, P, pre + 'conv1.', 3)
    cur, svd = y1, None
    if filt:
        cur = torch.empty_like(y1)
        svd = (dyn_fwd(y1[:, :h], P, pre + 'dyna.', 3, cur[:, :h]), dyn_fwd(y1[:, h:], P, pre + 'dyna_2.', 5, cur[:, h:]))
    r = torch.empty_like(cur)
    if _tlsc is not None:
        K.sf_local_affine(cur[:, :h], _box(K.sf_region_split(cur[:, :h], 1), H, W), P[pre + 'global_ap.fscale_h'], P[pre + 'global_ap.fscale_d'], 1.0, 1, r[:, :h])
        K.sf_local_affine(cur[:, h:], _box(K.sf_region_split(cur[:, h:], 2), H // 2, W // 2), P[pre + 'localap.h'], P[pre + 'localap.l'], 0.0, 2, r[:, h:])
        out, _ = conv_fwd(r, P, pre + 'conv2.', 3, act=False, res=x)
        return out, None
    mg = K.region_affine_fwd(cur[:, :h], P[pre + 'global_ap.fscale_h'], P[pre + 'global_ap.fscale_d'], 1.0, 1, r[:, :h])
    mp = K.region_affine_fwd(cur[:, h:], P[pre + 'localap.h'], P[pre + 'localap.l'], 0.0, 2, r[:, h:])
    out, sv2 = conv_fwd(r, P, pre + 'conv2.', 3, act=False, res=x)
    return out, (sv1, svd, cur, mg, mp, sv2)


def res_bwd(dout, P, pre, filt, saved, G):
    sv1, svd, cur, mg, mp, sv2 = saved
    N, C, H, W = cur.shape
    h = C // 2
    dout = dout.contiguous()
    dr = conv_bwd(dout, P, pre + 'conv2.', sv2, G)
    dcur = torch.empty_like(cur)
    G[pre + 'global_ap.fscale_h'], G[pre + 'global_ap.fscale_d'] = K.region_affine_bwd(
        dr[:, :h], cur[:, :h], P[pre + 'global_ap.fscale_h'], P[pre + 'global_ap.fscale_d'], 1.0, mg, 1, dcur[:, :h])
    G[pre + 'localap.h'], G[pre + 'localap.l'] = K.region_affine_bwd(
        dr[:, h:], cur[:, h:], P[pre + 'localap.h'], P[pre + 'localap.l'], 0.0, mp, 2, dcur[:, h:])
    dy1 = dcur
    if filt:
        dy1 = torch.empty_like(dcur)
        dyn_bwd(dcur[:, :h], P, pre + 'dyna.', 3, svd[0], dy1[:, :h], G)
        dyn_bwd(dcur[:, h:], P, pre + 'dyna_2.', 5, svd[1], dy1[:, h:], G)
    return conv_bwd(dy1, P, pre + 'conv1.', sv1, G, add_to_dx=dout)          # (+ the identity skip)


def blocks_fwd(x, P, pre, num_res):
    saved = []
    for r in range(num_res):
        x, sv = res_fwd(x, P, f'{pre}layers.{r}.', r == num_res - 1)
        saved.append(sv)
    return x, saved


def blocks_bwd(d, P, pre, num_res, saved, G):
    for r in reversed(range(num_res)):
        d = res_bwd(d, P, f'{pre}layers.{r}.', r == num_res - 1, saved[r], G)
    return d


# ------------------------------------------------------------------------------------------------- SCM / FAM (:200-223)
def scm_fwd(x, P, pre):
    a, s0 = conv_fwd(x, P, pre + 'main.0.', 3)
    b, s1 = conv_fwd(a, P, pre + 'main.1.', 1)
    c, s2 = conv_fwd(b, P, pre + 'main.2.', 3)
    d, s3 = conv_fwd(c, P, pre + 'main.3.', 1, act=False)
    y, mu, rs = K.instnorm_fwd(d, P[pre + 'main.4.weight'], P[pre + 'main.4.bias'])
    return y, (s0, s1, s2, s3, d, mu, rs)


def scm_bwd(dy, P, pre, saved, G):
    s0, s1, s2, s3, d, mu, rs = saved
    dd, G[pre + 'main.4.weight'], G[pre + 'main.4.bias'] = K.instnorm_bwd(dy.contiguous(), d, mu, rs, P[pre + 'main.4.weight'])
    dc = conv_bwd(dd, P, pre + 'main.3.', s3, G)
    db = conv_bwd(dc, P, pre + 'main.2.', s2, G)
    da = conv_bwd(db, P, pre + 'main.1.', s1, G)
    conv_bwd(da, P, pre + 'main.0.', s0, G, need_dx=False)                   # (the input image needs no gradient)


def cat_conv_fwd(a, b, P, pre, k, act):
    cat = K.concat2(a, b)
    y, sv = conv_fwd(cat, P, pre, k, act=act)
    return y, (sv, a.shape[1])


def cat_conv_bwd(dy, P, pre, saved, G):
    sv, ca = saved
    dcat = conv_bwd(dy, P, pre, sv, G)
    return dcat[:, :ca], dcat[:, ca:]


# ------------------------------------------------------------------------------------------------- SFNet.forward (:366-407)
def net_fwd(P, x, num_res, training=True, tlsc=None):
    """training=False: module.eval() -- BatchNorm2d on its running statistics, buffers untouched (InstanceNorm2d and the global pools are
    the same in both modes: the reference builds them without running statistics, sfnet_arch_utils.py:208, :108).
    tlsc = TLSC_BASE[mode[1]]: the mode[0] == 'test' network -- Gap / Patch_ap / SFconv pool with the box mean of that base size;
    inference only (nothing is kept for a backward pass)"""
    global _training, _tlsc
    if tlsc is not None and training:
        raise ValueError("SFNet mode 'test' is an inference network: training=False")
    _training, _tlsc = bool(training), tlsc
    try:
        return _net_fwd(P, x, num_res)
    finally:
        _training, _tlsc = True, None           # (the operator-level entry points default to the training pass)


def _net_fwd(P, x, num_res):
    x = x.contiguous()
    x_2 = K.subsample2(x)
    x_4 = K.subsample2(x_2)
    z2, sv_scm2 = scm_fwd(x_2, P, 'SCM2.')
    z4, sv_scm1 = scm_fwd(x_4, P, 'SCM1.')
    x_, sv_f0 = conv_fwd(x, P, 'feat_extract.0.', 3)
    res1, sv_e0 = blocks_fwd(x_, P, 'Encoder.0.', num_res)
    z, sv_f1 = conv_fwd(res1, P, 'feat_extract.1.', 3, stride=2)
    z, sv_fam2 = cat_conv_fwd(z, z2, P, 'FAM2.merge.', 3, False)
    res2, sv_e1 = blocks_fwd(z, P, 'Encoder.1.', num_res)
    z, sv_f2 = conv_fwd(res2, P, 'feat_extract.2.', 3, stride=2)
    z, sv_fam1 = cat_conv_fwd(z, z4, P, 'FAM1.merge.', 3, False)
    z, sv_e2 = blocks_fwd(z, P, 'Encoder.2.', num_res)
    zd0, sv_d0 = blocks_fwd(z, P, 'Decoder.0.', num_res)
    o4, sv_o0 = conv_fwd(zd0, P, 'ConvsOut.0.', 3, act=False, res=x_4)
    z, sv_f3 = convt_fwd(zd0, P, 'feat_extract.3.')
    z, sv_c0 = cat_conv_fwd(z, res2, P, 'Convs.0.', 1, True)
    zd1, sv_d1 = blocks_fwd(z, P, 'Decoder.1.', num_res)
    o2, sv_o1 = conv_fwd(zd1, P, 'ConvsOut.1.', 3, act=False, res=x_2)
    z, sv_f4 = convt_fwd(zd1, P, 'feat_extract.4.')
    z, sv_c1 = cat_conv_fwd(z, res1, P, 'Convs.1.', 1, True)
    zd2, sv_d2 = blocks_fwd(z, P, 'Decoder.2.', num_res)
    o1, sv_f5 = conv_fwd(zd2, P, 'feat_extract.5.', 3, act=False, res=x)
    saved = (num_res, sv_scm2, sv_scm1, sv_f0, sv_e0, sv_f1, sv_fam2, sv_e1, sv_f2, sv_fam1, sv_e2, sv_d0, sv_o0, sv_f3, sv_c0, sv_d1, sv_o1,
             sv_f4, sv_c1, sv_d2, sv_f5)
    return [o4, o2, o1], saved


def net_bwd(douts, P, saved, G=None):
    """douts = [d out 1/4, d out 1/2, d out full] -> {name: gradient} (the input image gets none: the reference trains on fixed inputs)"""
    (num_res, sv_scm2, sv_scm1, sv_f0, sv_e0, sv_f1, sv_fam2, sv_e1, sv_f2, sv_fam1, sv_e2, sv_d0, sv_o0, sv_f3, sv_c0, sv_d1, sv_o1, sv_f4,
     sv_c1, sv_d2, sv_f5) = saved
    G = {} if G is None else G
    d4, d2, d1 = (t.contiguous() for t in douts)
    with E.deferred_join():
        d = conv_bwd(d1, P, 'feat_extract.5.', sv_f5, G)
        d = blocks_bwd(d, P, 'Decoder.2.', num_res, sv_d2, G)
        dz, dres1 = cat_conv_bwd(d, P, 'Convs.1.', sv_c1, G)
        d = convt_bwd(dz, P, 'feat_extract.4.', sv_f4, G)
        d = K.add_(conv_bwd(d2, P, 'ConvsOut.1.', sv_o1, G), d)
        d = blocks_bwd(d, P, 'Decoder.1.', num_res, sv_d1, G)
        dz, dres2 = cat_conv_bwd(d, P, 'Convs.0.', sv_c0, G)
        d = convt_bwd(dz, P, 'feat_extract.3.', sv_f3, G)
        d = K.add_(conv_bwd(d4, P, 'ConvsOut.0.', sv_o0, G), d)
        d = blocks_bwd(d, P, 'Decoder.0.', num_res, sv_d0, G)
        d = blocks_bwd(d, P, 'Encoder.2.', num_res, sv_e2, G)
        dz, dz4 = cat_conv_bwd(d, P, 'FAM1.merge.', sv_fam1, G)
        d = conv_bwd(dz, P, 'feat_extract.2.', sv_f2, G, add_to_dx=dres2.contiguous())
        d = blocks_bwd(d, P, 'Encoder.1.', num_res, sv_e1, G)
        dz, dz2 = cat_conv_bwd(d, P, 'FAM2.merge.', sv_fam2, G)
        d = conv_bwd(dz, P, 'feat_extract.1.', sv_f1, G, add_to_dx=dres1.contiguous())
        d = blocks_bwd(d, P, 'Encoder.0.', num_res, sv_e0, G)
        conv_bwd(d, P, 'feat_extract.0.', sv_f0, G, need_dx=False)
        scm_bwd(dz4, P, 'SCM1.', sv_scm1, G)
        scm_bwd(dz2, P, 'SCM2.', sv_scm2, G)
    E.maybe_join()
    return G

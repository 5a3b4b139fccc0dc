"""GPU parity tests of the TRIPLE-plane (3 x bf16: h + m + l == x exactly) 3x3 convolution path -- the pre-split operands of
TDR_MATH=bx3, the reference's arithmetic (24-bit operands, fp32 range, no loss scale): csrc/tdr_conv_p16.hip (PF_TRI),
csrc/tdr_wgrad_p16.hip (NS = 3), through the C ABI.  Against torch-CPU fp32 / fp64 references of the same op, against the
fp32-tensor kernels they replace (bit-identical where the arithmetic is the same) and, for the MASA encoder as a whole, against
torch autograd.  Reference: models/archs/network_nafnet_guided_arch.py:44-59,110-143."""
import pytest
import torch
import torch.nn.functional as F

from test_hip_p16 import _encoder_params, _encoder_torch, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels
    prev = kernels.MATH
    kernels.set_math('bx3')
    yield kernels
    kernels.set_math(prev)


def test_triple_planes_hold_the_fp32_value_exactly(K):
    x = rnd(2, 32, 9, 13, seed=1, scale=3.0) * torch.logspace(-30, 20, 2 * 32 * 9 * 13, base=2.0).view(2, 32, 9, 13).cuda()
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, -1e-30]).cuda()
    x3 = K.p16_from_f32(x, fmt=K.FMT_BX3)
    torch.cuda.synchronize()
    assert x3.fmt == K.FMT_BX3 and x3.buf.numel() * 4 == 2 * 4 * 3 * 11 * 15 * 16
    buf = x3.buf.view(torch.bfloat16).view(2, 4, 3, 11, 15, 8).float()        # [N][C/8][plane][H+2][W+2][8]
    h = x.bfloat16()
    r = x - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    perm = lambda t: t.view(2, 4, 8, 9, 13).permute(0, 1, 3, 4, 2).float()
    assert torch.equal(buf[:, :, 0, 1:-1, 1:-1], perm(h))
    assert torch.equal(buf[:, :, 1, 1:-1, 1:-1], perm(m))
    assert torch.equal(buf[:, :, 2, 1:-1, 1:-1], perm(l))
    border = buf.clone()
    border[:, :, :, 1:-1, 1:-1] = 0
    assert border.abs().max().item() == 0.0
    # the three planes ARE the fp32 tensor: 8 + 8 + 8 significand bits on fp32's exponent, any magnitude
    assert torch.equal(x3.to_f32(), x)
    # exact zeros carry the sign bit in the head plane (the ReLU mask of the backward pass): -0.0
    heads = x3.buf.view(torch.int16).view(2, 4, 3, 11, 15, 8)[0, 0, 0, 1, 1:5, 0]
    assert heads[0].item() == -32768 and heads[1].item() == -32768 and heads[2].item() > 0 and heads[3].item() < 0


CASES = [  # N, Cin, Cout, H, W, residual, mask, relu
    (2, 32, 64, 32, 32, None, None, False),
    (1, 64, 128, 40, 64, 'f32', None, True),
    (2, 48, 32, 19, 45, 'p16', 'f32', False),
    (1, 16, 16, 8, 8, None, 'p16', False),
    (1, 128, 128, 64, 64, 'f32', 'f32', False),
    (1, 32, 24, 33, 31, None, None, True),
    (1, 256, 256, 16, 32, 'p16', 'p16', False),
]


@pytest.mark.parametrize('cfg', [0, 301, 302, 303, 304, 306, 307, 311, 321])
@pytest.mark.parametrize('case', CASES)
def test_conv3x3_triple_vs_torch_and_the_fp32_tensor_kernel(K, case, cfg):
    from textualdegremoval_amd import _lib
    N, Cin, Cout, H, W, res_kind, mask_kind, relu = case
    x = rnd(N, Cin, H, W, seed=2)
    w = rnd(Cout, Cin, 3, 3, seed=3, scale=0.05)
    b = rnd(Cout, seed=4)
    res = rnd(N, Cout, H, W, seed=5) if res_kind else None
    mask = rnd(N, Cout, H, W, seed=6) if mask_kind else None
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    assert wp.fmt == K.FMT_BX3
    x3 = K.p16_from_f32(x, fmt=K.FMT_BX3)
    res_in = K.p16_from_f32(res, fmt=K.FMT_BX3) if res_kind == 'p16' else res
    mask_in = K.p16_from_f32(mask, fmt=K.FMT_BX3) if mask_kind == 'p16' else mask
    _lib.load().tdr_conv3x3_p16_force_cfg(cfg)
    try:
        o32, o3 = K.conv3x3_p16(x3, wp, mp, Cout, bias=b, res=res_in, mask=mask_in, relu=relu, want32=True, want16=Cout % 16 == 0)
    finally:
        _lib.load().tdr_conv3x3_p16_force_cfg(0)
    ref = F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)
    if res is not None:
        ref = ref + res.cpu()
    if relu:
        ref = ref.clamp_min(0)
    if mask is not None:
        ref = torch.where(mask.cpu() > 0, ref, torch.zeros_like(ref))
    assert (o32.cpu() - ref).abs().max().item() < 1e-4
    # the fp32-tensor kernel of TDR_MATH=bx3 (conv_bx3_kernel<SCH_BX3>): same products, same accumulation order, and the planes
    # decode to the fp32 operands exactly -> bit-identical, residual in planes included
    old = K.conv_forward(x, wp, mp, Cout, 3, pad=1, bias=b, res=res, mask=mask, relu=relu)
    assert torch.equal(o32, old)
    if o3 is not None:
        assert torch.equal(o3.buf, K.p16_from_f32(o32, fmt=K.FMT_BX3).buf)
        assert torch.equal(o3.to_f32(), o32)


def test_triple_conv_has_no_window(K):
    """a power-of-two rescaling of the input commutes with the convolution bit for bit at gradient-sized magnitudes (2^-40: far
    below anything an fp16 plane could hold) -- the path needs no loss scale"""
    N, C, H, W = 1, 64, 24, 32
    x = rnd(N, C, H, W, seed=41)
    w = rnd(C, C, 3, 3, seed=42, scale=0.05)
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    a, _ = K.conv3x3_p16(K.p16_from_f32(x, fmt=K.FMT_BX3), wp, mp, C)
    b, _ = K.conv3x3_p16(K.p16_from_f32(x * 2.0 ** -40, fmt=K.FMT_BX3), wp, mp, C)
    c, _ = K.conv3x3_p16(K.p16_from_f32(x * 2.0 ** 40, fmt=K.FMT_BX3), wp, mp, C)
    assert torch.equal(a * 2.0 ** -40, b) and torch.equal(a * 2.0 ** 40, c)


def test_triple_mask_is_exactly_x_greater_than_zero(K):
    N, C, H, W = 1, 32, 16, 32
    vals = torch.tensor([1e-9, 0.0, -1e-9, 3e-8, 1.0, -0.0, 1e-30, 6e-8, -1.0, 2.0 ** -25, 2.0 ** -26, 1e-38])
    mask = vals.repeat((N * C * H * W + len(vals) - 1) // len(vals))[:N * C * H * W].view(N, C, H, W).cuda().contiguous()
    x = rnd(N, C, H, W, seed=31)
    w = rnd(C, C, 3, 3, seed=32, scale=0.05)
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    x3, m3 = K.p16_from_f32(x, fmt=K.FMT_BX3), K.p16_from_f32(mask, fmt=K.FMT_BX3)
    a, _ = K.conv3x3_p16(x3, wp, mp, C, mask=mask)
    b, _ = K.conv3x3_p16(x3, wp, mp, C, mask=m3)
    assert torch.equal(a, b)
    _, h3 = K.conv3x3_p16(x3, wp, mp, C, relu=True, want32=False, want16=True)
    h32, _ = K.conv3x3_p16(x3, wp, mp, C, relu=True)
    c, _ = K.conv3x3_p16(x3, wp, mp, C, mask=h3)
    d, _ = K.conv3x3_p16(x3, wp, mp, C, mask=h32)
    assert torch.equal(c, d)


@pytest.mark.parametrize('scale', [1.0, 2.0 ** -30])
@pytest.mark.parametrize('shape', [(1, 32, 32, 16, 32), (2, 64, 64, 32, 32), (1, 16, 48, 19, 45), (2, 128, 64, 24, 64),
                                   (1, 32, 32, 40, 33), (1, 64, 128, 7, 70), (2, 256, 256, 8, 32)])
def test_wgrad3x3_triple_vs_fp64(K, shape, scale):
    """scale 2^-30: output gradients of the size a real (unscaled) backward pass sees"""
    N, Cin, Cout, H, W = shape
    x = rnd(N, Cin, H, W, seed=7)
    d = rnd(N, Cout, H, W, seed=8) * scale
    x3, d3 = K.p16_from_f32(x, fmt=K.FMT_BX3), K.p16_from_f32(d, fmt=K.FMT_BX3)
    g, db = K.wgrad3x3_p16(x3, d3, want_db=True)
    xp = F.pad(x.double().cpu(), (1, 1, 1, 1))
    dd = d.double().cpu()
    ref = torch.empty(Cout, Cin, 3, 3, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = torch.einsum('nchw,nkhw->ck', dd, xp[:, :, ky:ky + H, kx:kx + W])
    assert (g[0].double().cpu() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert (db.double().cpu() - dd.sum((0, 2, 3))).abs().max().item() < 2e-6 * dd.sum((0, 2, 3)).abs().max().item()


def test_encoder_on_triples_equals_the_fp32_tensor_path(K, monkeypatch):
    """MASA encoder forward + backward on UNSCALED gradients of realistic size (1e-7): forward features bit-identical to the
    fp32-tensor kernels of the same arithmetic (the planes hold the residual stream exactly), parameter gradients within 2e-5 of
    their maximum of that path (the weight-gradient kernels sum in a different order) and within 2e-3 of torch autograd."""
    from textualdegremoval_amd import engine as E
    nf, cnt = 32, 2
    Pc = _encoder_params(nf, cnt)
    P = {k: v.cuda().contiguous() for k, v in Pc.items()}
    x = rnd(2, 3, 48, 64, seed=11)
    dfe = [rnd(2, nf * 2 ** l, 48 >> l, 64 >> l, seed=20 + l, scale=1e-7) for l in range(3)]

    def run(p16_on):
        monkeypatch.setattr(E, 'P16_ON', p16_on)
        monkeypatch.setattr(E, 'P16_MIN_C', 32)
        assert not K.GRAD_SCALED
        feats, saved = E.encoder_fwd(x, P, 'masa_enc.', [cnt, cnt, cnt], levels=3)
        used = any(bl and isinstance(bl[0][0], K.P16) and bl[0][0].fmt == K.FMT_BX3 for _, _, bl in saved)
        G = {}
        E.encoder_bwd([d.clone() for d in dfe], P, 'masa_enc.', [cnt, cnt, cnt], saved, G)
        torch.cuda.synchronize()
        return feats, G, used

    f_new, g_new, used_new = run(True)
    f_old, g_old, used_old = run(False)
    assert used_new and not used_old
    Pt = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
    f_ref = _encoder_torch(x.cpu(), Pt, cnt)
    sum((f * d.cpu()).sum() for f, d in zip(f_ref, dfe)).backward()
    for a, b, c in zip(f_new, f_old, f_ref):
        assert torch.equal(a, b)
        assert (a.cpu() - c.detach()).abs().max().item() < 1e-4
    worst_ref = worst_old = 0.0
    for k in Pc:
        gr = Pt[k].grad
        sc = gr.abs().max().item()
        worst_ref = max(worst_ref, (g_new[k].cpu().view_as(gr) - gr).abs().max().item() / sc)
        worst_old = max(worst_old, (g_new[k] - g_old[k].view_as(g_new[k])).abs().max().item() / sc)
    print('triple encoder: worst vs autograd', worst_ref, 'vs fp32-tensor path', worst_old)
    assert worst_ref < 2e-3, worst_ref
    assert worst_old < 2e-5, worst_old

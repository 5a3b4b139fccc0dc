"""GPU parity tests of the pre-split ("P16") 3x3 convolution path (csrc/tdr_conv_p16.hip, csrc/tdr_wgrad_p16.hip) through the
C ABI: against a torch-CPU fp32 / fp64 reference of the same op, against the fp32-tensor kernels it replaces (bit-identical
where the arithmetic is the same), and -- for the MASA encoder as a whole -- against the oracle's Encoder
(oracle/nafnet_ref_oracle.py, pinned by tests/test_oracle_golden.py) forward and backward.
Reference: models/archs/network_nafnet_guided_arch.py:44-59,110-143."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels
    prev = kernels.MATH
    kernels.set_math('hx2')
    yield kernels
    kernels.set_math(prev)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def test_p16_image_is_the_fp16_pair_with_a_zero_border(K):
    x = rnd(2, 32, 9, 13, seed=1, scale=3.0)
    x16 = K.p16_from_f32(x)
    torch.cuda.synchronize()
    buf = x16.buf.view(torch.float16).view(2, 4, 2, 11, 15, 8).float()       # [N][C/8][plane][H+2][W+2][8]
    h = x.half()
    m = (x - h.float()).half()
    want_h = h.view(2, 4, 8, 9, 13).permute(0, 1, 3, 4, 2).float()
    want_m = m.view(2, 4, 8, 9, 13).permute(0, 1, 3, 4, 2).float()
    assert torch.equal(buf[:, :, 0, 1:-1, 1:-1], want_h)
    assert torch.equal(buf[:, :, 1, 1:-1, 1:-1], want_m)
    border = buf.clone()
    border[:, :, :, 1:-1, 1:-1] = 0
    assert border.abs().max().item() == 0.0
    # head + residual carries >= 22 significant bits
    assert (x16.to_f32() - x).abs().max().item() <= 2.0 ** -21 * x.abs().max().item()


CASES = [  # N, Cin, Cout, H, W, residual, mask, relu
    (2, 32, 64, 32, 32, None, None, False),
    (1, 64, 128, 40, 64, 'f32', None, True),
    (2, 48, 32, 19, 45, 'p16', 'f32', False),
    (1, 16, 16, 8, 8, None, 'p16', False),
    (1, 128, 128, 64, 64, 'f32', 'f32', False),
    (1, 32, 24, 33, 31, None, None, True),
    (1, 256, 256, 16, 32, 'p16', 'p16', False),
]


@pytest.mark.parametrize('cfg', [0, 3, 16, 17, 19, 22])
@pytest.mark.parametrize('case', CASES)
def test_conv3x3_p16_vs_torch_and_the_fp32_tensor_kernel(K, case, cfg):
    from textualdegremoval_amd import _lib
    N, Cin, Cout, H, W, res_kind, mask_kind, relu = case
    x = rnd(N, Cin, H, W, seed=2)
    w = rnd(Cout, Cin, 3, 3, seed=3, scale=0.05)
    b = rnd(Cout, seed=4)
    res = rnd(N, Cout, H, W, seed=5) if res_kind else None
    mask = rnd(N, Cout, H, W, seed=6) if mask_kind else None
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    x16 = K.p16_from_f32(x)
    res_in = K.p16_from_f32(res) if res_kind == 'p16' else res
    mask_in = K.p16_from_f32(mask) if mask_kind == 'p16' else mask
    _lib.load().tdr_conv3x3_p16_force_cfg(cfg)
    try:
        o32, o16 = K.conv3x3_p16(x16, wp, mp, Cout, bias=b, res=res_in, mask=mask_in, relu=relu, want32=True, want16=Cout % 16 == 0)
    finally:
        _lib.load().tdr_conv3x3_p16_force_cfg(0)
    # (1) torch CPU fp32 reference of the op: <= 1e-4 max-abs on O(1) tensors (measured ~1e-6)
    ref = F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)
    if res is not None:
        ref = ref + res.cpu()
    if relu:
        ref = ref.clamp_min(0)
    if mask is not None:
        ref = torch.where(mask.cpu() > 0, ref, torch.zeros_like(ref))
    assert (o32.cpu() - ref).abs().max().item() < 1e-4
    # (2) the fp32-tensor kernel with the same arithmetic and accumulation order: bit-identical (the P16 residual decodes
    # to head + residual, so that case is compared against the decoded tensor)
    res_ref = res_in.to_f32() if res_kind == 'p16' else res
    old = K.conv_forward(x, wp, mp, Cout, 3, pad=1, bias=b, res=res_ref, mask=mask, relu=relu)
    assert torch.equal(o32, old)
    # (3) the P16 output is the split of the fp32 output, border included
    if o16 is not None:
        assert torch.equal(o16.buf, K.p16_from_f32(o32).buf)


def test_p16_mask_is_exactly_x_greater_than_zero(K):
    """ReLU masks are read from the pair planes.  fp16 heads underflow below 2^-25, so "x > 0" is carried by the SIGN BIT of the head
    with exact zeros stored as -0.0: tiny positives (head +0) pass, exact zeros and negatives do not -- the same decisions as the
    fp32 mask, element for element (a plain rn_f16(x) > 0 test dropped a few gradient terms per tensor at configs[1]'s size)."""
    N, C, H, W = 1, 32, 16, 32
    vals = torch.tensor([1e-9, 0.0, -1e-9, 3e-8, 1.0, -0.0, 1e-30, 6e-8, -1.0, 2.0 ** -25, 2.0 ** -26, 1e-38])
    mask = vals.repeat((N * C * H * W + len(vals) - 1) // len(vals))[:N * C * H * W].view(N, C, H, W).cuda().contiguous()
    x = rnd(N, C, H, W, seed=31)
    w = rnd(C, C, 3, 3, seed=32, scale=0.05)
    wp, mp, *_ = K.pack_weights(w, K.PACK_FWD)
    x16, m16 = K.p16_from_f32(x), K.p16_from_f32(mask)
    a, _ = K.conv3x3_p16(x16, wp, mp, C, mask=mask)
    b, _ = K.conv3x3_p16(x16, wp, mp, C, mask=m16)
    assert torch.equal(a, b)
    assert ((a != 0) == (mask > 0)).all() or (a[mask > 0] == 0).float().mean().item() < 1e-3     # kept exactly where mask > 0
    # and a ReLU output written by the kernel itself carries the same information
    _, h16 = K.conv3x3_p16(x16, wp, mp, C, relu=True, want32=False, want16=True)
    h32, _ = K.conv3x3_p16(x16, wp, mp, C, relu=True)
    c, _ = K.conv3x3_p16(x16, wp, mp, C, mask=h16)
    d, _ = K.conv3x3_p16(x16, wp, mp, C, mask=h32)
    assert torch.equal(c, d)


@pytest.mark.parametrize('shape', [(1, 32, 32, 16, 32), (2, 64, 64, 32, 32), (1, 16, 48, 19, 45), (2, 128, 64, 24, 64),
                                   (1, 32, 32, 40, 33), (1, 64, 128, 7, 70), (2, 256, 256, 8, 32)])
def test_wgrad3x3_p16_vs_fp64(K, shape):
    N, Cin, Cout, H, W = shape
    x = rnd(N, Cin, H, W, seed=7)
    d = rnd(N, Cout, H, W, seed=8)
    x16, d16 = K.p16_from_f32(x), K.p16_from_f32(d)
    g, db = K.wgrad3x3_p16(x16, d16, want_db=True)
    xp = F.pad(x.double().cpu(), (1, 1, 1, 1))
    dd = d.double().cpu()
    ref = torch.empty(Cout, Cin, 3, 3, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = torch.einsum('nchw,nkhw->ck', dd, xp[:, :, ky:ky + H, kx:kx + W])
    # fp32-class products and fp32 accumulation over N*H*W pixels: 2e-6 of the tensor maximum (measured 1 - 3e-7)
    assert (g[0].double().cpu() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert (db.double().cpu() - dd.sum((0, 2, 3))).abs().max().item() < 2e-6 * dd.sum((0, 2, 3)).abs().max().item()


def _encoder_params(nf, cnt, seed=0):
    g = torch.Generator().manual_seed(seed)
    P = {}
    cin = 3
    for k in range(1, 4):
        c = nf * 2 ** (k - 1)
        P[f'masa_enc.conv_L{k}.weight'] = torch.randn(c, cin, 3, 3, generator=g) * (0.6 / (cin * 9) ** 0.5)
        P[f'masa_enc.conv_L{k}.bias'] = torch.randn(c, generator=g) * 0.05
        for i in range(cnt):
            for j in (1, 2):
                P[f'masa_enc.blk_L{k}.{i}.conv{j}.weight'] = torch.randn(c, c, 3, 3, generator=g) * (0.6 / (c * 9) ** 0.5)
                P[f'masa_enc.blk_L{k}.{i}.conv{j}.bias'] = torch.randn(c, generator=g) * 0.05
        cin = c
    return P


def _encoder_torch(x, P, cnt):
    """the reference Encoder (:110-143) restated with autograd for this test: conv_L + ReLU, ResidualBlocks conv-ReLU-conv + skip"""
    feats = []
    for k in range(1, 4):
        x = F.relu(F.conv2d(x, P[f'masa_enc.conv_L{k}.weight'], P[f'masa_enc.conv_L{k}.bias'], stride=1 if k == 1 else 2, padding=1))
        for i in range(cnt):
            bp = f'masa_enc.blk_L{k}.{i}.'
            h = F.relu(F.conv2d(x, P[bp + 'conv1.weight'], P[bp + 'conv1.bias'], padding=1))
            x = F.conv2d(h, P[bp + 'conv2.weight'], P[bp + 'conv2.bias'], padding=1) + x
        feats.append(x)
    return feats


def test_encoder_on_pairs_matches_autograd_and_the_fp32_tensor_path(K, monkeypatch):
    """MASA encoder (nf 32: levels of 32 / 64 / 128 channels) forward + backward inside a loss-scaled step: the P16 path
    against torch autograd on the CPU (outputs 1e-4, parameter gradients 2e-3 of their maximum -- the bars of the
    whole-network tests) and against the fp32-tensor path of the same engine (1e-5 / 1e-4: same products, the pair-encoded
    residual stream aside)."""
    from textualdegremoval_amd import engine as E
    nf, cnt, S = 32, 2, 2.0 ** 10
    Pc = _encoder_params(nf, cnt)
    P = {k: v.cuda().contiguous() for k, v in Pc.items()}
    x = rnd(2, 3, 48, 64, seed=11)
    dfe = [rnd(2, nf * 2 ** l, 48 >> l, 64 >> l, seed=20 + l, scale=1e-3) for l in range(3)]

    def run(p16_on):
        monkeypatch.setattr(E, 'P16_ON', p16_on)
        monkeypatch.setattr(E, 'P16_MIN_C', 32)
        prev = K.set_grad_scaled(True)
        try:
            feats, saved = E.encoder_fwd(x, P, 'masa_enc.', [cnt, cnt, cnt], levels=3)
            used = any(bl and isinstance(bl[0][0], K.P16) for _, _, bl in saved)
            G = {}
            E.encoder_bwd([d * S for d in dfe], P, 'masa_enc.', [cnt, cnt, cnt], saved, G)
        finally:
            K.set_grad_scaled(prev)
        torch.cuda.synchronize()
        return feats, {k: v / S for k, v in G.items()}, used

    f_new, g_new, used_new = run(True)
    f_old, g_old, used_old = run(False)
    assert used_new and not used_old
    Pt = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
    f_ref = _encoder_torch(x.cpu(), Pt, cnt)
    sum((f * d.cpu()).sum() for f, d in zip(f_ref, dfe)).backward()
    for a, b, c in zip(f_new, f_old, f_ref):
        assert (a.cpu() - c.detach()).abs().max().item() < 1e-4
        assert (a - b).abs().max().item() < 1e-5
    worst_ref = worst_old = 0.0
    for k in Pc:
        gr = Pt[k].grad
        sc = gr.abs().max().item()
        worst_ref = max(worst_ref, (g_new[k].cpu().view_as(gr) - gr).abs().max().item() / sc)
        worst_old = max(worst_old, (g_new[k] - g_old[k]).abs().max().item() / sc)
    assert worst_ref < 2e-3, worst_ref
    assert worst_old < 1e-4, worst_old

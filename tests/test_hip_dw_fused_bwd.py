"""One-pass backward of the depthwise 3x3 stencils (csrc/tdr_dwsg.hip dwsg_bwd_fused_kernel: SimpleGate of NAFNet-ref,
network_nafnet_guided_arch.py:170-187; GDFN gate and qkv_dwconv of Restormer-ref, network_restormer_guided_arch.py:236-260)
against a torch fp32 autograd reference of the same op and against the two-pass kernels (TDR_DWSG_TWO_PASS=1), on shapes
that exercise every neighbour path: rows inside one wave, several row strips per wave, rows spanning 2 / 4 waves (LDS edge
exchange), widths that are not a power of two, heights that are not a multiple of the strip length."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 6, 64, 64), (1, 4, 37, 96), (2, 3, 130, 256), (1, 5, 100, 384), (1, 4, 70, 512), (1, 2, 40, 1024),
          (1, 3, 9, 8), (2, 2, 256, 128), (1, 2, 33, 1028)]      # the last one is wider than one column block: two-pass kernels


def _ref(kind, t, w, b, dg):
    t = t.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    b = None if b is None else b.clone().requires_grad_(True)
    u = F.conv2d(t, w, b, padding=1, groups=t.shape[1])
    c = t.shape[1] // 2
    if kind == 'mul':
        g = u[:, :c] * u[:, c:]
    elif kind == 'gelu':
        g = F.gelu(u[:, :c]) * u[:, c:]
    else:
        g = u
    g.backward(dg)
    return t.grad, w.grad, None if b is None else b.grad


def _run(kind, K, dg, t, w, b, dgb=None):
    if kind == 'mul':
        return K.dwsg_bwd(dg, t, w, b, dgb, 0.5) if dgb is not None else K.dwsg_bwd(dg, t, w, b)
    if kind == 'gelu':
        return K.dwgelu_bwd(dg, t, w, b)
    return K.dwconv_bwd(dg, t, w, want_db=True)


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('kind', ['mul', 'gelu', 'none'])
def test_one_pass_backward_matches_autograd_and_two_pass(kind, shape):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * 1000 + W)
    t = torch.randn(N, 2 * C, H, W, generator=g).cuda()
    w = (torch.randn(2 * C, 1, 3, 3, generator=g) * 0.4).cuda()
    b = (torch.randn(2 * C, generator=g) * 0.2).cuda() if kind != 'none' else None
    dg = torch.randn(N, C if kind != 'none' else 2 * C, H, W, generator=g).cuda()
    want = _ref(kind, t, w, b, dg)
    got = _run(kind, K, dg, t, w, b)
    os.environ['TDR_DWSG_TWO_PASS'] = '1'
    try:
        two = _run(kind, K, dg, t, w, b)
    finally:
        os.environ.pop('TDR_DWSG_TWO_PASS')
    scale = [1.0, (H * W * N) ** 0.5, (H * W * N) ** 0.5]
    for i, (a, r, s2) in enumerate(zip(got, want, two)):
        if r is None:
            continue
        tol = 2e-5 * scale[i] * max(1.0, float(r.abs().max()) / scale[i])
        assert float((a.view_as(r) - r).abs().max()) < tol, (i, float((a.view_as(r) - r).abs().max()), tol)
        assert float((a - s2).abs().max()) < tol, ('two-pass', i)
    # dt is computed with the same arithmetic in both versions
    assert torch.equal(got[0], two[0]) or float((got[0] - two[0]).abs().max()) < 1e-5


def test_one_pass_backward_with_pooled_gradient_bias():
    """the SCA branch's pooled gradient enters as a per-plane constant on dg (engine.naf_bwd)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    N, C, H, W = 2, 8, 48, 512
    g = torch.Generator().manual_seed(3)
    t = torch.randn(N, 2 * C, H, W, generator=g).cuda()
    w = (torch.randn(2 * C, 1, 3, 3, generator=g) * 0.4).cuda()
    b = (torch.randn(2 * C, generator=g) * 0.2).cuda()
    dg = torch.randn(N, C, H, W, generator=g).cuda()
    dgb = torch.randn(N, C, generator=g).cuda()
    want = _ref('mul', t, w, b, dg + 0.5 * dgb[:, :, None, None])
    got = _run('mul', K, dg, t, w, b, dgb)
    assert float((got[0] - want[0]).abs().max()) < 5e-5
    assert float((got[1].view_as(want[1]) - want[1]).abs().max()) < 2e-5 * (H * W * N) ** 0.5 * 4


@pytest.mark.parametrize('shape', [(2, 6, 64, 64), (1, 4, 37, 96), (1, 5, 100, 384), (1, 4, 70, 512), (1, 2, 40, 1024)])
@pytest.mark.parametrize('mult,relu,bias', [(1, True, True), (1, False, False), (2, True, True), (2, True, False), (2, False, True)])
@pytest.mark.parametrize('Kk', [3, 5])
def test_msfn_depthwise_on_the_stencils(shape, mult, relu, bias, Kk):
    """DRSformer-ref MSFN (network_drsformer_guided_arch.py:226-253): relu(dwconv3x3(x)) / relu(dwconv5x5(x)) with one (groups =
    channels) or two (groups = channels / 2) inputs per output, through kernels.dwk_fwd / dwk_bwd: K = 3 runs on tdr_dwsg.hip,
    the K = 5 backward on tdr_dwk.hip's one-pass kernel"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    N, C, H, W = shape
    Cout = 2 * C
    g = torch.Generator().manual_seed(H + W + mult)
    x = torch.randn(N, Cout * mult, H, W, generator=g).cuda()
    w = (torch.randn(Cout, mult, Kk, Kk, generator=g) * (0.4 if Kk == 3 else 0.25)).cuda()
    b = (torch.randn(Cout, generator=g) * 0.3).cuda() if bias else None
    dy = torch.randn(N, Cout, H, W, generator=g).cuda()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=Kk // 2, groups=Cout)
    if relu:
        yr = F.relu(yr)
    yr.backward(dy)
    y = K.dwk_fwd(x, w, b, relu=relu)
    assert float((y - yr).abs().max()) < 2e-5
    dx, dw, db = K.dwk_bwd(dy, y if relu else None, x, w, want_db=bias)
    assert float((dx - xr.grad).abs().max()) < 5e-5
    s = (N * H * W) ** 0.5
    assert float((dw - wr.grad).abs().max()) < 2e-5 * s * max(1.0, float(wr.grad.abs().max()) / s)
    if bias:
        assert float((db - br.grad).abs().max()) < 2e-5 * s * max(1.0, float(br.grad.abs().max()) / s)
    if Kk != 3:
        return
    K.DWK_GENERIC = True                             # the LDS-tiled generic kernels of tdr_dwk.hip agree
    try:
        y2 = K.dwk_fwd(x, w, b, relu=relu)
        dx2, dw2, _ = K.dwk_bwd(dy, y2 if relu else None, x, w, want_db=bias)
    finally:
        K.DWK_GENERIC = False
    assert float((y - y2).abs().max()) < 2e-5 and float((dx - dx2).abs().max()) < 5e-5
    assert float((dw - dw2).abs().max()) < 2e-5 * s * max(1.0, float(wr.grad.abs().max()) / s)


@pytest.mark.parametrize('Kk', [3, 5])
def test_msfn_second_stage_writes_into_and_reads_from_channel_slices(Kk):
    """z1 / z2 of MSFN go straight into the concatenated buffer, and their gradients / ReLU masks are read from channel slices of
    wider tensors (drsformer_engine.ffn_fwd / ffn_bwd): per-image strides through the pair stencil (K = 3) and the 5x5 kernels"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    N, h, H, W = 2, 7, 40, 64
    g = torch.Generator().manual_seed(Kk)
    x = torch.randn(N, 2 * h, H, W, generator=g).cuda()
    w = (torch.randn(h, 2, Kk, Kk, generator=g) * 0.3).cuda()
    cat = torch.full((N, 2 * h + 3, H, W), 7.0, device='cuda')
    dcat = torch.randn(N, 2 * h + 3, H, W, generator=g).cuda()
    y = K.dwk_fwd(x, w, None, relu=True, out=cat[:, 2:2 + h])
    assert y.data_ptr() == cat[:, 2:2 + h].data_ptr()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, wr, None, padding=Kk // 2, groups=h))
    assert float((cat[:, 2:2 + h] - yr).abs().max()) < 2e-5
    assert float((cat[:, :2] - 7).abs().max()) == 0 and float((cat[:, 2 + h:] - 7).abs().max()) == 0     # neighbours untouched
    yr.backward(dcat[:, 1:1 + h])
    dx, dw, _ = K.dwk_bwd(dcat[:, 1:1 + h], cat[:, 2:2 + h], x, w)
    assert float((dx - xr.grad).abs().max()) < 5e-5
    s = (N * H * W) ** 0.5
    assert float((dw - wr.grad).abs().max()) < 2e-5 * s * max(1.0, float(wr.grad.abs().max()) / s)


@pytest.mark.parametrize('relu,bias', [(True, True), (False, False)])
def test_plain_depthwise_3x3_with_split_halves(relu, bias):
    """MSFN's cross-concatenation in place (drsformer_engine.ffn_fwd): the 3x3 stencil stores planes [0, h) and [h, 2h) of its
    output into channel slices of two different buffers, and the backward reads dout / the ReLU mask split the same way"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    N, h, H, W = 2, 7, 37, 96
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, 2 * h, H, W, generator=g).cuda()
    w = (torch.randn(2 * h, 1, 3, 3, generator=g) * 0.4).cuda()
    b = (torch.randn(2 * h, generator=g) * 0.3).cuda() if bias else None
    x1 = torch.full((N, 2 * h, H, W), 5.0, device='cuda')
    x2 = torch.full((N, 2 * h, H, W), 6.0, device='cuda')
    ya, yb = K.dwk_fwd(x, w, b, relu=relu, out=(x1[:, :h], x2[:, h:]))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=1, groups=2 * h)
    if relu:
        yr = F.relu(yr)
    assert float((x1[:, :h] - yr[:, :h]).abs().max()) < 2e-5 and float((x2[:, h:] - yr[:, h:]).abs().max()) < 2e-5
    assert float((x1[:, h:] - 5).abs().max()) == 0 and float((x2[:, :h] - 6).abs().max()) == 0
    d1 = torch.randn(N, 2 * h, H, W, generator=g).cuda()
    d2 = torch.randn(N, 2 * h, H, W, generator=g).cuda()
    yr.backward(torch.cat([d1[:, h:], d2[:, :h]], dim=1))
    dx, dw, db = K.dwk_bwd((d1[:, h:], d2[:, :h]), (ya, yb) if relu else None, x, w, want_db=bias)
    assert float((dx - xr.grad).abs().max()) < 5e-5
    s = (N * H * W) ** 0.5
    assert float((dw - wr.grad).abs().max()) < 2e-5 * s * max(1.0, float(wr.grad.abs().max()) / s)
    if bias:
        assert float((db - br.grad).abs().max()) < 2e-5 * s * max(1.0, float(br.grad.abs().max()) / s)


@pytest.mark.parametrize('Kk', [5, 7])
def test_depthwise_backward_accumulates_into_dx(Kk):
    """dx_out += (MSFN: the 5x5 branch adds its input gradient onto the 3x3 branch's): inside the one-pass 5x5 kernel, through a
    temporary + add for the kernels that cannot (K = 7 here)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    N, C, H, W = 2, 6, 33, 64
    g = torch.Generator().manual_seed(Kk)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(C, 1, Kk, Kk, generator=g) * 0.2).cuda()
    dy = torch.randn(N, C, H, W, generator=g).cuda()
    base = torch.randn(N, C, H, W, generator=g).cuda()
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=Kk // 2, groups=C).backward(dy)
    dx = base.clone()
    out, _, _ = K.dwk_bwd(dy, None, x, w, dx_out=dx, accumulate=True)
    assert out.data_ptr() == dx.data_ptr()
    assert float((dx - (base + xr.grad)).abs().max()) < 5e-5

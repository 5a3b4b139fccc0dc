"""GPU parity tests of the HIP kernels (through the C ABI) against torch-CPU
fp32 references / the oracle, on seeded inputs.  Tolerances are stated per
test; the target of the path is 1e-4 max-abs on O(1) fp32 tensors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nafnet_ref_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', params=['bx3', 'f32', 'hx2'])
def K(request):
    """every kernel test runs under both matrix-core arithmetic modes of the dense convolutions
    (split-bf16 'bx3' = product default, exact fp32 MFMA 'f32')."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels
    prev, prev1 = kernels.MATH, kernels.WGRAD_1X1_BX3
    kernels.set_math(request.param)
    kernels.WGRAD_1X1_BX3 = True          # exercise the split 1x1 weight-gradient kernel as well
    yield kernels
    kernels.set_math(prev)
    kernels.WGRAD_1X1_BX3 = prev1


def dev(t):
    return t.cuda().contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def rel(a, b):
    b = b.detach().cpu().double()
    return maxdiff(a, b) / max(b.abs().max().item(), 1e-12)


# ------------------------------------------------------------------ conv fwd
CONV_CASES = [
    # N, Cin, Cout, H, W, KH, stride, dil, pad
    (2, 32, 64, 64, 64, 1, 1, 1, 0),
    (1, 128, 128, 32, 32, 1, 1, 1, 0),
    (2, 8, 16, 8, 8, 1, 1, 1, 0),
    (3, 48, 24, 16, 24, 1, 1, 1, 0),
    (4, 160, 160, 64, 64, 1, 1, 1, 0),
    (2, 256, 96, 24, 40, 1, 1, 1, 0),      # >= 4 K stages: the 16-byte-staged 1x1 kernel (hx2), ragged tiles
    (1, 204, 72, 16, 36, 1, 1, 1, 0),      # ... partial last octet / group
    (1, 512, 128, 32, 32, 1, 1, 1, 0),
    (4, 256, 512, 64, 64, 1, 1, 1, 0),     # ... 128 x 128 tiles, one round of workgroups
    (1, 3, 8, 32, 48, 3, 1, 1, 1),
    (2, 32, 32, 64, 64, 3, 1, 1, 1),
    (2, 64, 64, 16, 16, 3, 1, 1, 1),
    (1, 256, 256, 8, 8, 3, 1, 1, 1),
    (4, 128, 128, 64, 64, 3, 1, 1, 1),
    (8, 32, 32, 256, 256, 3, 1, 1, 1),     # 2048 workgroups: the many-round variant without the weight-fragment ring
    (1, 20, 12, 13, 21, 3, 1, 1, 1),
    (2, 16, 32, 32, 32, 3, 2, 1, 1),
    (1, 32, 64, 64, 96, 3, 2, 1, 1),
    (2, 16, 32, 32, 32, 2, 2, 1, 0),
    (1, 64, 128, 16, 16, 2, 2, 1, 0),
    (2, 16, 4, 12, 12, 3, 1, 2, 2),
    (2, 16, 4, 12, 12, 3, 1, 3, 3),
    (2, 24, 64, 15, 15, 3, 1, 1, 0),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_plain(K, case):
    N, Cin, Cout, H, W, KH, st, dl, pd = case
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, KH, KH, seed=2, scale=1.0 / (Cin * KH * KH) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.3)
    ref = F.conv2d(x, w, b, stride=st, padding=pd, dilation=dl)
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD, math='f32' if dl != 1 else None)   # dilation: search path, exact only
    out = K.conv_forward(dev(x), wp, mp, Cout, KH, stride=st, dil=dl, pad=pd, bias=dev(b))
    assert out.shape == ref.shape
    assert maxdiff(out, ref) < 2e-5


@pytest.mark.parametrize('cfg', [1, 2, 3, 4, 5])
def test_conv1x1_every_tile_configuration(K, cfg):
    """every (waves, tiles) configuration of the 1x1 kernels -- under hx2 with >= 4 K stages that is the float4-staged
    kernel in all five shapes (128x128, 64x256, 64x128, 32x256, 256x64), otherwise the 16-channel pipeline."""
    from textualdegremoval_amd import _lib
    lib = _lib.load()
    N, Cin, Cout, H, W = 2, 520, 200, 20, 36
    x = rnd(N, Cin, H, W, seed=1); w = rnd(Cout, Cin, 1, 1, seed=2, scale=1.0 / Cin ** 0.5); b = rnd(Cout, seed=3, scale=0.3)
    res = rnd(N, Cout, H, W, seed=4)
    ref = F.conv2d(x, w, b) + res
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD)
    lib.tdr_conv_force_cfg(1, cfg)
    try:
        out = K.conv_forward(dev(x), wp, mp, Cout, 1, bias=dev(b), res=dev(res))
    finally:
        lib.tdr_conv_force_cfg(1, 0)
    assert maxdiff(out, ref) < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('cfg', [1, 2, 3, 4])
def test_conv3x3_every_tile_configuration(K, cfg):
    """128x256, 64x256, 64x128, 128x128 tiles of the 3x3 kernel (the one-m-tile-per-wave shapes run the 9-slot
    weight-fragment ring at this launch size)."""
    from textualdegremoval_amd import _lib
    lib = _lib.load()
    N, Cin, Cout, H, W = 2, 72, 136, 24, 40
    x = rnd(N, Cin, H, W, seed=1); w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / (9 * Cin) ** 0.5); b = rnd(Cout, seed=3, scale=0.3)
    ref = F.conv2d(x, w, b, padding=1)
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD)
    lib.tdr_conv_force_cfg(3, cfg)
    try:
        out = K.conv_forward(dev(x), wp, mp, Cout, 3, pad=1, bias=dev(b))
    finally:
        lib.tdr_conv_force_cfg(3, 0)
    assert maxdiff(out, ref) < 2e-5 * max(1.0, ref.abs().max().item())


def test_conv_forward_epilogue_chain(K):
    N, Cin, Cout, H, W = 2, 32, 48, 24, 40
    x = rnd(N, Cin, H, W, seed=1); w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.06); b = rnd(Cout, seed=3)
    ks = rnd(N, Cin, seed=4) + 1.0; sc = rnd(N, Cout, seed=5); b2 = rnd(N, Cout, seed=6)
    res = rnd(N, Cout, H, W, seed=7); mask = rnd(N, Cout, H, W, seed=8)
    ref = F.conv2d(x * ks.view(N, Cin, 1, 1), w, b, padding=1)
    ref = ref * sc.view(N, Cout, 1, 1) + 0.25 * b2.view(N, Cout, 1, 1) + res
    ref = torch.relu(ref) * (mask > 0)
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD)
    out = K.conv_forward(dev(x), wp, mp, Cout, 3, pad=1, kscale=dev(ks), bias=dev(b), scale=dev(sc), bias2=dev(b2),
                         bias2_mul=0.25, res=dev(res), mask=dev(mask), relu=True)
    assert maxdiff(out, ref) < 3e-5


def test_conv_forward_strided_views(K):
    """input / residual / output addressed as channel slices of bigger buffers."""
    N, C, H, W = 3, 16, 16, 16
    big = rnd(N, 2 * C, H, W, seed=1); w = rnd(C, C, 1, 1, seed=2, scale=0.25); b = rnd(C, seed=3)
    resb = rnd(N, 3 * C, H, W, seed=4)
    ref = F.conv2d(big[:, C:], w, b) + resb[:, C:2 * C]
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD)
    outb = torch.zeros(N, 2 * C, H, W, device='cuda')
    K.conv_forward(dev(big)[:, C:], wp, mp, C, 1, bias=dev(b), res=dev(resb)[:, C:2 * C], out=outb[:, :C])
    assert maxdiff(outb[:, :C], ref) < 2e-5
    assert outb[:, C:].abs().max().item() == 0.0


@pytest.mark.parametrize('C', [32, 256])
def test_conv_forward_gate_and_gatebwd(K, C):
    N, H, W = 2, 16, 32
    t4 = rnd(N, 2 * C, H, W, seed=1); w = rnd(C, C, 1, 1, seed=2, scale=0.2); b = rnd(C, seed=3)
    gam = rnd(C, seed=4); y = rnd(N, C, H, W, seed=5)
    ref = y + F.conv2d(t4[:, :C] * t4[:, C:], w, b) * gam.view(1, C, 1, 1)
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD)
    out = K.conv_forward(dev(t4), wp, mp, C, 1, gate=True, bias=dev(b), scale=dev(gam), res=dev(y))
    assert maxdiff(out, ref) < 2e-5 * max(1.0, ref.abs().max().item())
    # GATEBWD: dt4 from dout
    dout = rnd(N, C, H, W, seed=6)
    dg2 = F.conv_transpose2d(dout * gam.view(1, C, 1, 1), w)
    ref_dt4 = torch.cat([dg2 * t4[:, C:], dg2 * t4[:, :C]], 1)
    wpd, mpd, *_ = K.pack_weights(dev(w), K.PACK_DGRAD_S1)
    dt4 = K.conv_forward(dev(dout), wpd, mpd, C, 1, epi=K.EPI_GATEBWD, kscale=dev(gam), aux=dev(t4))
    assert maxdiff(dt4, ref_dt4) < 2e-5 * max(1.0, ref_dt4.abs().max().item())


def test_conv_forward_pixelshuffle_with_skip(K):
    N, C, H, W = 2, 32, 8, 16
    x = rnd(N, C, H, W, seed=1); w = rnd(2 * C, C, 1, 1, seed=2, scale=0.2); skip = rnd(N, C // 2, 2 * H, 2 * W, seed=3)
    ref = F.pixel_shuffle(F.conv2d(x, w), 2) + skip
    wp, mp, *_ = K.pack_weights(dev(w), K.PACK_FWD)
    out = K.conv_forward(dev(x), wp, mp, 2 * C, 1, epi=K.EPI_PSHUF, res=dev(skip))
    assert maxdiff(out, ref) < 2e-5


@pytest.mark.parametrize('KH,st,pd,Cin,Cout,H,W', [(1, 1, 0, 32, 64, 16, 16), (3, 1, 1, 16, 24, 20, 28),
                                                    (3, 1, 1, 64, 64, 32, 32), (2, 2, 0, 16, 32, 32, 32),
                                                    (3, 2, 1, 16, 32, 32, 48), (3, 2, 1, 8, 16, 16, 16)])
def test_conv_data_gradient(K, KH, st, pd, Cin, Cout, H, W):
    N = 2
    x = rnd(N, Cin, H, W, seed=1).requires_grad_(True)
    w = rnd(Cout, Cin, KH, KH, seed=2, scale=0.1)
    y = F.conv2d(x, w, stride=st, padding=pd)
    go = rnd(*y.shape, seed=3)
    y.backward(go)
    from textualdegremoval_amd import engine as E
    dx, dw, db = E.conv_bwd(dev(go), dev(x.detach()), dev(w), st, pd)
    assert maxdiff(dx, x.grad) < 3e-5
    ref_dw = torch.autograd.grad(F.conv2d(x, w.requires_grad_(True), stride=st, padding=pd), w, go)[0]
    assert rel(dw, ref_dw) < 2e-5
    assert rel(db, go.sum(dim=(0, 2, 3))) < 2e-5


# ------------------------------------------------------------------ multi-tensor weight packing
def test_pack_multi_equals_single(K):
    """the one-launch multi-tensor pack of a step (forward layout: one thread per lane of a fragment row, all taps, 16-byte loads) writes
    the same bytes as the per-weight pack kernels, for every layout the engine asks for -- incl. ragged channel counts, Cin < 8 and a
    weight that is a 4-byte-aligned view of a larger buffer"""
    shapes = [(512, 256, 1), (64, 64, 3), (32, 3, 3), (24, 20, 3), (64, 32, 2), (40, 72, 1), (128, 128, 3), (3, 32, 3)]
    ws = [dev(rnd(co, ci, kh, kh, seed=10 + i)) for i, (co, ci, kh) in enumerate(shapes)]
    flat = dev(rnd(1 + 48 * 40 * 9, seed=99))
    ws.append(flat[1:].view(48, 40, 3, 3))                         # data_ptr only 4-byte aligned
    modes = {1: [K.PACK_FWD, K.PACK_DGRAD_S1], 3: [K.PACK_FWD, K.PACK_DGRAD_S1], 2: [K.PACK_FWD, K.PACK_DGRAD_2X2S2]}
    singles = [(w, m, K.pack_weights(w, m)[0]) for w in ws for m in modes[w.shape[2]]]
    plan = K.PackPlan()
    prev = K.set_pack_plan(plan)
    try:
        for w, m, _ in singles:
            K.pack_weights(w, m)                                   # records (and packs on the spot)
        for e in plan.entries.values():
            e[3].buf.fill_(float('nan'))
        plan.invalidate()
        plan.run()                                                 # the multi-tensor launch
        torch.cuda.synchronize()
        for w, m, ref in singles:
            got = K.pack_weights(w, m)[0]
            assert got.fmt == ref.fmt
            assert torch.equal(got.buf.view(torch.int32), ref.buf.view(torch.int32)), (tuple(w.shape), m)
    finally:
        K.set_pack_plan(prev)


# ------------------------------------------------------------------ wgrad
@pytest.mark.parametrize('N,Cin,Cout,H,W,KH,st,pd', [
    (2, 128, 128, 32, 32, 1, 1, 0), (2, 32, 64, 64, 64, 1, 1, 0), (2, 32, 32, 48, 64, 1, 1, 0),
    (3, 64, 128, 16, 16, 1, 1, 0), (2, 256, 512, 8, 8, 1, 1, 0), (1, 8, 16, 8, 8, 1, 1, 0),
    (2, 32, 32, 64, 64, 3, 1, 1), (2, 64, 64, 32, 32, 3, 1, 1), (1, 3, 32, 64, 64, 3, 1, 1),
    (2, 32, 3, 32, 32, 3, 1, 1), (1, 128, 128, 8, 8, 3, 1, 1), (2, 16, 32, 32, 32, 3, 2, 1),
    (2, 16, 32, 32, 32, 2, 2, 0), (1, 8, 8, 13, 17, 3, 1, 1),
    # 3x3 stride 2 (csrc/tdr_wgrad_s2.hip under hx2): Cin <= 32 / > 32 variants, several 32-column strips, blocks that start
    # mid-column and cross a column (OH = 20: 8 tiles per block), ragged channel blocks and a ragged last strip (OW = 40)
    (2, 32, 64, 128, 128, 3, 2, 1), (1, 64, 128, 64, 64, 3, 2, 1), (2, 128, 256, 32, 32, 3, 2, 1),
    (1, 72, 80, 40, 80, 3, 2, 1), (3, 8, 16, 64, 64, 3, 2, 1), (1, 40, 24, 16, 24, 3, 2, 1),
    # wide 1x1 layers: ragged channel tiles, pixel counts that do not divide into the blocks' tiles, three images
    (2, 256, 512, 64, 64, 1, 1, 0), (1, 96, 72, 40, 40, 1, 1, 0), (3, 64, 64, 40, 40, 1, 1, 0), (2, 192, 64, 32, 40, 1, 1, 0),
    # 2x2 stride 2 (the `downs`), same kernel family
    (2, 32, 64, 128, 128, 2, 2, 0), (1, 64, 128, 64, 64, 2, 2, 0), (1, 72, 80, 40, 80, 2, 2, 0), (2, 128, 256, 32, 32, 2, 2, 0)])
def test_conv_wgrad(K, N, Cin, Cout, H, W, KH, st, pd):
    x = rnd(N, Cin, H, W, seed=1)
    w = torch.zeros(Cout, Cin, KH, KH, requires_grad=True)
    y = F.conv2d(x, w, stride=st, padding=pd)
    go = rnd(*y.shape, seed=2)
    ref = torch.autograd.grad(y, w, go)[0]
    # under 'hx2' the O(1) operands are declared fp16-range, so the 2-way fp16 split variant of the kernel is the one tested
    g = K.conv_wgrad(dev(x), dev(go), Cout, Cin, KH, stride=st, pad=pd, fp16_range=True)
    assert rel(g.view_as(ref), ref) < 2e-5


def test_conv_wgrad_stride2_bias_and_per_image(K):
    """bias gradient riding on the stride-2 weight gradient, and the per-image groups (reference conv_L2 .. conv_L5,
    models/archs/network_nafnet_guided_arch.py:122-128)"""
    N, Cin, Cout, H, W = 3, 48, 96, 48, 64
    x = rnd(N, Cin, H, W, seed=3); go = rnd(N, Cout, H // 2, W // 2, seed=4)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(x, w, stride=2, padding=1), w, go)[0]
    g, db = K.conv_wgrad(dev(x), dev(go), Cout, Cin, 3, stride=2, pad=1, want_db=True, fp16_range=True)
    assert rel(g.view_as(ref), ref) < 2e-5
    assert rel(db, go.sum(dim=(0, 2, 3))) < 2e-5
    gi = K.conv_wgrad(dev(x), dev(go), Cout, Cin, 3, stride=2, pad=1, per_image=True, fp16_range=True)
    assert rel(gi.sum(0).view_as(ref), ref) < 2e-5
    for n in range(N):
        w1 = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
        r1 = torch.autograd.grad(F.conv2d(x[n:n + 1], w1, stride=2, padding=1), w1, go[n:n + 1])[0]
        assert rel(gi[n].view_as(r1), r1) < 2e-5


def test_conv_wgrad_wide_gate_bias_and_per_image(K):
    """SimpleGate operand, fused bias gradient and per-image groups on the wide 1x1 weight-gradient kernel (conv3 / conv5 of a
    NAFBlock, reference models/archs/network_nafnet_guided_arch.py:226-238)"""
    N, C, H, W = 3, 96, 32, 40
    t4 = rnd(N, 2 * C, H, W, seed=5); go = rnd(N, 80, H, W, seed=6)
    g2 = t4[:, :C] * t4[:, C:]
    ref = torch.einsum('nohw,nihw->noi', go, g2)
    g = K.conv_wgrad(dev(t4), dev(go), 80, C, 1, gate=True, per_image=True, fp16_range=True)
    assert rel(g.view(N, 80, C), ref) < 2e-5
    gs, db = K.conv_wgrad(dev(t4), dev(go), 80, C, 1, gate=True, want_db=True, fp16_range=True)
    assert rel(gs.view(80, C), ref.sum(0)) < 2e-5
    assert rel(db, go.sum(dim=(0, 2, 3))) < 2e-5
    x = rnd(N, C, H, W, seed=7)
    g3, db3 = K.conv_wgrad(dev(x), dev(go), 80, C, 1, want_db=True, fp16_range=True)
    assert rel(g3.view(80, C), torch.einsum('nohw,nihw->oi', go, x)) < 2e-5
    assert rel(db3, go.sum(dim=(0, 2, 3))) < 2e-5


def test_conv_wgrad_gate_and_per_image(K):
    N, C, H, W = 3, 32, 16, 16
    t4 = rnd(N, 2 * C, H, W, seed=1); go = rnd(N, C, H, W, seed=2)
    g2 = t4[:, :C] * t4[:, C:]
    ref = torch.einsum('nohw,nihw->noi', go, g2)
    g = K.conv_wgrad(dev(t4), dev(go), C, C, 1, gate=True, per_image=True, fp16_range=True)
    assert rel(g.view(N, C, C), ref) < 2e-5
    gs = K.conv_wgrad(dev(t4), dev(go), C, C, 1, gate=True, fp16_range=True)
    assert rel(gs.view(C, C), ref.sum(0)) < 2e-5


# ------------------------------------------------------------------ LayerNorm2d
@pytest.mark.parametrize('C,H,W', [(8, 9, 10), (32, 16, 16), (48, 8, 24), (64, 32, 32), (128, 16, 16), (256, 8, 8),
                                   (512, 8, 8), (1024, 4, 8)])
def test_layernorm2d(K, C, H, W):
    N = 2
    x = rnd(N, C, H, W, seed=1).requires_grad_(True)
    w = (rnd(C, seed=2) * 0.3 + 1).requires_grad_(True); b = (rnd(C, seed=3) * 0.2).requires_grad_(True)
    go = rnd(N, C, H, W, seed=4); add = rnd(N, C, H, W, seed=5)
    y = O.layernorm2d(x, w, b, 1e-6)
    y.backward(go)
    yk, mu, rs = K.layernorm2d_fwd(dev(x.detach()), dev(w.detach()), dev(b.detach()), 1e-6)
    assert maxdiff(yk, y) < 2e-5
    gx, gw, gb = K.layernorm2d_bwd(dev(go), dev(x.detach()), mu, rs, dev(w.detach()), add=dev(add))
    assert maxdiff(gx, x.grad + add) < 5e-5
    assert rel(gw, w.grad) < 2e-5 and rel(gb, b.grad) < 2e-5


# ------------------------------------------------------------------ dw3x3 + SimpleGate
@pytest.mark.parametrize('C,H,W', [(8, 8, 8), (16, 24, 40), (32, 64, 64), (4, 128, 512)])
def test_dwsg(K, C, H, W):
    N = 2
    t = rnd(N, 2 * C, H, W, seed=1).requires_grad_(True)
    w = (rnd(2 * C, 1, 3, 3, seed=2) * 0.3).requires_grad_(True); b = (rnd(2 * C, seed=3) * 0.2).requires_grad_(True)
    u = F.conv2d(t, w, b, padding=1, groups=2 * C)
    g = u[:, :C] * u[:, C:]
    go = rnd(N, C, H, W, seed=4)
    g.backward(go)
    gk, pooled = K.dwsg_fwd(dev(t.detach()), dev(w.detach()), dev(b.detach()))
    assert maxdiff(gk, g) < 2e-5
    assert maxdiff(pooled, g.mean(dim=(2, 3))) < 2e-5
    dt, dw, db = K.dwsg_bwd(dev(go), dev(t.detach()), dev(w.detach()), dev(b.detach()))
    assert maxdiff(dt, t.grad) < 3e-5
    assert rel(dw, w.grad) < 3e-5 and rel(db, b.grad) < 3e-5


# ------------------------------------------------------------------ SCA chain / small kernels
def test_sca_and_param_chains(K):
    N, C, H, W = 3, 32, 8, 8
    g = rnd(N, C, H, W, seed=1); dy = rnd(N, C, H, W, seed=2)
    w3 = (rnd(C, C, 1, 1, seed=3) * 0.2).requires_grad_(True); b3 = rnd(C, seed=4).requires_grad_(True)
    beta = rnd(1, C, 1, 1, seed=5).requires_grad_(True)
    wsca = (rnd(C, C, 1, 1, seed=6) * 0.2).requires_grad_(True); bsca = rnd(C, seed=7).requires_grad_(True)
    pooled = g.mean(dim=(2, 3)).requires_grad_(True)
    s = F.conv2d(pooled.view(N, C, 1, 1), wsca, bsca)
    out = F.conv2d(g * s, w3, b3) * beta
    out.backward(dy)
    sk = K.sca_fwd(dev(pooled.detach()), dev(wsca.detach()), dev(bsca.detach()))
    assert maxdiff(sk, s.view(N, C)) < 1e-5
    G3 = torch.einsum('nohw,nihw->noi', dy, g)
    S3 = dy.sum(dim=(0, 2, 3))
    r = K.sca_bwd(dev(G3), dev(S3), dev(w3.detach()), dev(b3.detach()), dev(beta.detach().view(-1)), sk,
                  dev(pooled.detach()), dev(wsca.detach()))
    dw3, db3, dbeta, dwsca, dbsca, dpooled = r
    assert rel(dw3, w3.grad) < 3e-5 and rel(db3, b3.grad) < 3e-5 and rel(dbeta, beta.grad) < 3e-5
    assert rel(dwsca, wsca.grad) < 3e-5 and rel(dbsca, bsca.grad) < 3e-5 and rel(dpooled, pooled.grad) < 3e-5
    # gamma chain
    w5 = (rnd(C, C, 1, 1, seed=8) * 0.2).requires_grad_(True); b5 = rnd(C, seed=9).requires_grad_(True)
    gam = rnd(1, C, 1, 1, seed=10).requires_grad_(True)
    (F.conv2d(g, w5, b5) * gam).backward(dy)
    G5 = G3.sum(0)
    dw5, db5, dgam = K.scaled_conv_param_grads(dev(G5), dev(S3), dev(w5.detach()), dev(b5.detach()), dev(gam.detach().view(-1)))
    assert rel(dw5, w5.grad.view(C, C)) < 3e-5 and rel(db5, b5.grad) < 3e-5 and rel(dgam, gam.grad.view(-1)) < 3e-5


def test_glue_kernels(K):
    N, C, H, W = 2, 8, 12, 16
    a = rnd(N, C, H, W, seed=1); b = rnd(N, 2 * C, H, W, seed=2)
    cat = K.concat2(dev(a), dev(b))
    assert maxdiff(cat, torch.cat([a, b], 1)) == 0
    assert maxdiff(K.slice_channels(cat, C, 2 * C), b[:, :C]) == 0
    da = dev(a).clone()
    K.add_(da, dev(a))
    assert maxdiff(da, 2 * a) == 0
    assert maxdiff(K.channel_sum(dev(b)), b.sum(dim=(0, 2, 3))) < 1e-4
    assert maxdiff(K.pixel_unshuffle2(dev(a)), F.pixel_unshuffle(a, 2)) == 0
    assert maxdiff(K.pad_crop(dev(a), 16, 24), F.pad(a, (0, 8, 0, 4))) == 0
    assert maxdiff(K.pad_crop(dev(a), 8, 12), a[:, :, :8, :12]) == 0
    assert maxdiff(K.relu_bwd(dev(a), dev(b[:, :C].contiguous())), a * (b[:, :C] > 0)) == 0
    p = rnd(N, 3, 20, 24, seed=3); t = rnd(N, 3, 20, 24, seed=4)
    pr = p.clone().requires_grad_(True)
    l = 0.7 * (pr - t).abs().mean(); l.backward()
    loss, dp = K.l1_loss(dev(p), dev(t), 0.7)
    assert abs(loss.item() - l.item()) < 1e-6 and maxdiff(dp, pr.grad) < 1e-9


# ------------------------------------------------------------------ MASA pieces vs golden (reference outputs)
def gold(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_lr_blocks_fwd_bwd(K):
    N, C, H, W, py, px, k = 2, 6, 16, 24, 2, 3, 8
    f = rnd(N, C, H, W, seed=1).requires_grad_(True)
    ref = O.lr_blocks(f, py, px, k, k)
    go = rnd(*ref.shape, seed=2)
    ref.backward(go)
    blk = K.lr_blocks_fwd(dev(f.detach()), py, px, k, k)
    assert maxdiff(blk.view(N, py * px, C, k + 2, k + 2), ref) == 0
    d = K.lr_blocks_bwd(dev(go.reshape(N * py * px, C, k + 2, k + 2)), N, C, H, W, py, px, k, k)
    assert maxdiff(d, f.grad) < 1e-5


@pytest.mark.parametrize('s', [1, 2, 4])
def test_transfer_vs_reference_golden(K, s):
    g = gold('masa_ops')
    fea = T(g[f'tr{s}_fea']); att = T(g[f'tr{s}_att']); idx = T(g[f'tr{s}_idx'])
    B, C = fea.shape[0], fea.shape[1]
    # each "image" holds one block: py=px=1, block start (0,0), side 15
    y1 = torch.zeros(B, dtype=torch.int32, device='cuda'); x1 = torch.zeros_like(y1)
    ia = dev(idx.int().view(B, 64)); sa = dev(att.view(B, 64))
    out = K.transfer_fwd(dev(fea), y1, x1, ia, sa, 1, 1, 8, 15, s)
    assert maxdiff(out, T(g[f'tr{s}_out'])) < 1e-5
    dfeat = torch.zeros_like(dev(fea)); datt = torch.zeros(B, 64, device='cuda')
    K.transfer_bwd(dev(T(g[f'tr{s}_go'])), dev(fea), y1, x1, ia, sa, 1, 1, 8, 15, s, dfeat, datt)
    assert maxdiff(dfeat, T(g[f'tr{s}_gfea'])) < 2e-5
    assert maxdiff(datt.view(B, 1, 8, 8), T(g[f'tr{s}_gatt'])) < 5e-5


def test_fine_search_vs_reference_golden(K):
    g = gold('masa_ops')
    lr = T(g['so_lr']); rf = T(g['so_ref'])
    B, C = lr.shape[0], lr.shape[1]
    wp, mp, per_b = K.pack_patches(dev(lr), 1, 8, 8, 1, 1, 0)
    dots = K.conv_forward(dev(rf), wp, mp, 64, 3, pad=0, wp_ns=per_b)
    iq = K.patch_inv_norm(dev(lr), 8, 8); ik = K.patch_inv_norm(dev(rf), 13, 13)
    idx, att = K.fine_argmax(dots, iq, ik, B, 64, 169)
    assert np.array_equal(idx.cpu().numpy().reshape(B, 8, 8), g['so_idx'])
    assert maxdiff(att.view(B, 8, 8), T(g['so_val'])) < 2e-6
    dl, dr = K.fine_search_bwd(dev(T(g['so_go']).reshape(B, 64)), att, idx, dev(lr), dev(rf), iq, ik, 8, 15)
    assert maxdiff(dl, T(g['so_glr'])) < 2e-5
    assert maxdiff(dr, T(g['so_gref'])) < 2e-5


def test_coarse_search_and_box_vs_reference_golden(K):
    g = gold('masa_ops')
    lrp = T(g['cs_lr']); ref = T(g['cs_ref'])               # [N,P,C,10,10], [N,C,12,12]
    N, P, C = lrp.shape[0], lrp.shape[1], lrp.shape[2]
    Hr = ref.shape[2]
    lrb = dev(lrp.reshape(N * P, C, 10, 10)); r4 = dev(ref)
    dots = torch.empty(3, N, P, Hr, Hr, device='cuda'); iq = torch.empty(3, N * P, device='cuda'); ik = torch.empty(3, N, Hr * Hr, device='cuda')
    for di, d in enumerate([1, 2, 3]):
        wp, mp, per_b = K.pack_patches(lrb, P, 1, 1, 1, d, 5 - d)
        K.conv_forward(r4, wp, mp, P, 3, dil=d, pad=d, wp_ns=per_b, out=dots[di])
        K.patch_inv_norm(lrb, 1, 1, dil=d, off=5 - d, out=iq[di])
        K.patch_inv_norm(r4, Hr, Hr, dil=d, pad=d, out=ik[di])
    index, y1, x1 = K.coarse_argmax_box(dots, iq, ik, N, P, Hr, Hr, 13)
    assert np.array_equal(index.cpu().numpy().reshape(N, P), g['cs_idx'])
    ii = torch.from_numpy(g['cs_idx']).long()
    assert np.array_equal(x1.cpu().numpy().reshape(N, P), O.box_start(ii % Hr, Hr, 13).numpy())
    assert np.array_equal(y1.cpu().numpy().reshape(N, P), O.box_start(ii // Hr, Hr, 13).numpy())
    # wrap-aware gather + scatter adjoint (Hr=12 < 15 -> negative starts)
    blk = K.gather_ref_block(r4, y1, x1, P, 15, 1)
    refb = O.gather_ref_block(ref, O.box_start(ii // Hr, Hr, 13), O.box_start(ii % Hr, Hr, 13), 15, 1)
    assert maxdiff(blk, refb) == 0
    rr = ref.clone().requires_grad_(True)
    rb = O.gather_ref_block(rr, O.box_start(ii // Hr, Hr, 13), O.box_start(ii % Hr, Hr, 13), 15, 1)
    go = rnd(*rb.shape, seed=5); rb.backward(go)
    df = torch.zeros_like(r4)
    K.scatter_ref_block(dev(go), df, y1, x1, P, 15)
    assert maxdiff(df, rr.grad) < 1e-5


@pytest.mark.parametrize('B,C,H,W,dil,pad', [(2, 70, 10, 12, 1, 0),      # 160 positions: wave-per-position kernel
                                               (3, 70, 20, 24, 1, 0),      # 1188 positions: position-major kernel
                                               (2, 96, 24, 24, 2, 2), (2, 40, 24, 24, 3, 3)])
def test_patch_inv_norm_both_kernels(K, B, C, H, W, dil, pad):
    """1 / ||3x3 (dilated) patch over all channels||, zero outside the map (network_nafnet_guided_arch.py:515-536)."""
    x = rnd(B, C, H, W, seed=11)
    OH, OW = H + 2 * pad - 2 * dil, W + 2 * pad - 2 * dil
    ss = F.conv2d((x * x).sum(1, keepdim=True), torch.ones(1, 1, 3, 3), dilation=dil, padding=pad)[:, 0]
    ref = 1.0 / ss.sqrt().clamp_min(1e-12)
    out = K.patch_inv_norm(dev(x), OH, OW, dil=dil, pad=pad)
    assert out.shape == ref.shape
    assert rel(out, ref) < 2e-6


# ------------------------------------------------------------------ optimiser
def test_clip_and_adamw_match_torch(K):
    from textualdegremoval_amd.optim import FusedClipAdamW
    shapes = [(5000,), (7, 3, 3, 3), (1, 16, 1, 1), (33,), (4097,)]
    ps = [torch.nn.Parameter(rnd(*s, seed=i)) for i, s in enumerate(shapes)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_ref = torch.optim.AdamW([{'params': ref[:2], 'lr': 2e-4}, {'params': ref[2:], 'lr': 1e-4}], lr=2e-4,
                                weight_decay=1e-4, betas=(0.9, 0.999))
    gp = [torch.nn.Parameter(p.detach().cuda()) for p in ps]
    opt = FusedClipAdamW([{'params': gp[:2], 'lr': 2e-4}, {'params': gp[2:], 'lr': 1e-4}], lr=2e-4, weight_decay=1e-4,
                         betas=(0.9, 0.999), max_norm=0.01, use_grad_clip=True)
    for step in range(3):
        for i, (a, b) in enumerate(zip(ref, gp)):
            gr = rnd(*a.shape, seed=100 + 10 * step + i) * 0.01
            a.grad = gr.clone(); b.grad = gr.cuda()
        torch.nn.utils.clip_grad_norm_(ref, 0.01)
        opt_ref.step(); opt.step()
    for a, b in zip(ref, gp):
        assert maxdiff(b, a) < 1e-6

"""GPU parity of the frozen ViT window matcher (through the C ABI) against the oracle and the golden vectors recorded
from the reference classes.  Tolerances: tokens 1e-4 max-abs (O(1) values), correlation 1e-5, indices exact."""
import os

import numpy as np
import pytest
import torch

from oracle import dino_oracle as D
from tests.test_dino_oracle_golden import images, match_inputs

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', params=['bx3', 'f32', 'hx2'])
def matcher(request):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.dino import DinoMatcher
    prev = K.MATH
    K.set_math(request.param)
    sd = D.synth_vit_params(32, 2, 2, seed=11)
    yield DinoMatcher(sd, torch.device('cuda'), heads=2, linear_math=request.param), sd
    K.set_math(prev)


def test_helper_kernels(matcher):
    from textualdegremoval_amd import kernels as K
    import torch.nn.functional as F
    x = images(2, 48, 48, seed=3)
    y = K.resize_bilinear(x.cuda(), 56, 56).cpu()
    assert (y - F.interpolate(x, size=(56, 56), mode='bilinear')).abs().max().item() < 1e-6
    ref = images(2, 96, 96, seed=4)
    w, N = K.unfold_windows(ref.cuda(), 48, 12)
    un = F.unfold(ref, kernel_size=(48, 48), stride=(12, 12)).transpose(-1, -2).contiguous().view(2 * N, 3, 48, 48)
    assert N == 25 and torch.equal(w.cpu(), un)


@pytest.mark.parametrize('tag,shape', [('sq56', (2, 56, 56)), ('rect70x42', (1, 70, 42)), ('sq140', (1, 140, 140))])
def test_patch_tokens(matcher, tag, shape):
    m, sd = matcher
    gold = np.load(os.path.join(GOLDEN, 'dino_vit_e32_d2.npz'))
    B, H, W = shape
    x = images(B, H, W, seed=100 + H)
    tok, T = m.tokens(x.cuda())
    t = tok.reshape(B, tok.shape[1], -1)[:, :, 1:T + 1].transpose(1, 2).cpu()        # [B, T, D]
    ref = D.vit_patch_tokens(sd, x, heads=2)
    assert (t - ref).abs().max().item() < 1e-4
    assert np.abs(t.numpy() - gold[f'tokens_{tag}']).max() < 1e-4


def test_flat_layout_gives_the_same_tokens(matcher):
    m, sd = matcher
    x = images(3, 70, 42, seed=5)
    a, T = m.tokens(x.cuda())
    b, Tb = m.tokens(x.cuda(), flat=True)
    assert T == Tb and a.shape == b.shape
    A, Bt = a.reshape(3, a.shape[1], -1)[:, :, :T + 1], b.reshape(3, b.shape[1], -1)[:, :, :T + 1]
    assert torch.equal(A, Bt)


def test_window_match(matcher):
    m, sd = matcher
    gold = np.load(os.path.join(GOLDEN, 'dino_vit_e32_d2.npz'))
    lq, ref = match_inputs()
    ref_in, idx, corr = m.match(lq.cuda(), ref.cuda())
    o_ref_in, o_idx, o_corr = D.match_reference_window(sd, lq, ref, heads=2)
    assert np.array_equal(idx.cpu().numpy(), gold['match_index']) and torch.equal(idx.cpu().long(), o_idx)
    assert (corr.cpu() - o_corr).abs().max().item() < 1e-5
    assert torch.equal(ref_in.cpu(), o_ref_in)


def test_vit_b14_head_dim_64(matcher):
    """one block of real ViT-B geometry (D=768, 12 heads of 64) on a 70x70 image: attention kernel HD=64 path"""
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.dino import DinoMatcher
    sd = D.synth_vit_params(768, 1, 12, seed=5)
    m = DinoMatcher(sd, torch.device('cuda'), heads=12, linear_math=K.MATH)
    x = images(1, 70, 70, seed=9)
    tok, T = m.tokens(x.cuda())
    t = tok.reshape(1, 768, -1)[:, :, 1:T + 1].transpose(1, 2).cpu()
    ref = D.vit_patch_tokens(sd, x, heads=12)
    assert (t - ref).abs().max().item() < 2e-4


def test_tok16_kernels_against_torch():
    """csrc/tdr_tok16.hip piece by piece against torch on the same fp16-rounded operands (fp32 accumulation on both sides: the bars
    are summation-order noise, 1e-3 of the output scale, plus half an fp16 ulp where the output is fp16)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    x = r(3, 70, 45)
    assert torch.equal(K.transpose_f32(x), x.transpose(1, 2).contiguous())
    P, D = 2 * 160 + 37, 768                                               # a ragged last row tile
    t, w, b = r(P, D) * 3 + 1, r(D), r(D)
    ref = torch.nn.functional.layer_norm(t, (D,), w, b, 1e-6)
    assert (K.tok_layernorm(t, w, b, 1e-6, out_f16=False) - ref).abs().max().item() < 2e-5
    h16 = K.tok_layernorm(t, w, b, 1e-6)
    assert h16.dtype == torch.float16 and (h16.float() - ref).abs().max().item() < 1e-5 + 2 ** -11 * ref.abs().max().item()
    for N, Kd in ((2304, 768), (768, 3072)):
        x16, w16, bias, ls = (r(P, Kd) * 0.5).half(), (r(N, Kd) * 0.05).half(), r(N), r(N)
        acc = x16.float() @ w16.float().t() + bias
        bar = 1e-3 * acc.abs().max().item()
        y0, y1 = K.tok16_gemm(x16, w16, bias), K.tok16_gemm(x16, w16, bias, epi=1)
        assert (y0.float() - acc).abs().max().item() < bar + 2 ** -11 * acc.abs().max().item()
        assert (y1.float() - torch.nn.functional.gelu(acc)).abs().max().item() < bar + 2 ** -11 * acc.abs().max().item()
        res = r(P, N)
        want = res + ls * acc
        got = K.tok16_gemm(x16, w16, bias, epi=2, res=res.clone(), ls=ls)
        assert (got - want).abs().max().item() < 1e-3 * want.abs().max().item()
        assert (K.tok16_gemm(x16, w16, None, epi=2, res=res.clone()) - (res + acc - bias)).abs().max().item() < 1e-3 * want.abs().max().item()
    B, heads, T1, LD = 3, 12, 257 + 64, 352                              # 5 full key tiles + 1 key, a partly filled query block
    qkv = (r(B * LD, 3 * 768) * 0.7).half()
    out = K.tok16_attention(qkv, B, heads, 0.125, T1).float().view(B, LD, heads, 64)
    q, k, v = (qkv.float().view(B, LD, 3, heads, 64)[:, :T1, i].permute(0, 2, 1, 3) for i in range(3))
    want = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3)
    assert (out[:, :T1] - want).abs().max().item() < 3e-3 * want.abs().max().item()        # P is rounded to fp16 before P V
    assert torch.equal(out[:, T1:], torch.zeros_like(out[:, T1:]))


def test_single_product_linears_keep_every_match_decision():
    """DinoMatcher(linear_math='h1'): the frozen Linears AND the attention (tdr_attention_fwd_math math code 3) on ONE fp16 MFMA product.
    The sub-graph's only output is the window index, so
    the bar is bit-exact indices: on the reference golden, and -- full ViT-B/14 geometry, structured 128x128 crops against 256x256
    references at several offsets / seeds (25 windows each) -- against the 2-way split arithmetic, with the similarity margin between
    the best and the second-best window reported."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.dino import DinoMatcher, random_vit_b14_state_dict
    prev = K.MATH
    K.set_math('hx2')
    try:
        qkv = torch.randn(3, 3 * 128, 9, 32, generator=torch.Generator().manual_seed(5)).cuda()      # head dim 64, 257 of 288 columns
        a2, a3 = K.attention_fwd(qkv, 2, 0.125, 257), K.attention_fwd(qkv, 2, 0.125, 257, single_product=True)
        err = (a2 - a3).view(3, 128, -1)[..., :257].abs().max().item()
        assert 0 < err < 4e-3 * a2.abs().max().item(), err                    # a different (plain fp16) arithmetic, and close
        assert torch.equal(a3.view(3, 128, -1)[..., 257:], torch.zeros_like(a3.view(3, 128, -1)[..., 257:]))
        sd = D.synth_vit_params(32, 2, 2, seed=11)
        m1 = DinoMatcher(sd, torch.device('cuda'), heads=2, linear_math='h1')
        gold = np.load(os.path.join(GOLDEN, 'dino_vit_e32_d2.npz'))
        lq, ref = match_inputs()
        ref_in, idx, corr = m1.match(lq.cuda(), ref.cuda())
        o_ref_in, o_idx, o_corr = D.match_reference_window(sd, lq, ref, heads=2)
        assert np.array_equal(idx.cpu().numpy(), gold['match_index']) and torch.equal(ref_in.cpu(), o_ref_in)
        assert (corr.cpu() - o_corr).abs().max().item() < 5e-3
        sdb = random_vit_b14_state_dict(seed=1, depth=12)
        # the default is the fp32-faithful split (round 4: 'h1' is opt-in -- its arg-max is only pinned on random-init weights)
        assert DinoMatcher(sdb, torch.device('cuda')).linear_math is None and not DinoMatcher(sdb, torch.device('cuda')).tok16
        a, b = DinoMatcher(sdb, torch.device('cuda'), linear_math='hx2'), DinoMatcher(sdb, torch.device('cuda'), linear_math='h1')
        assert b.linear_math == 'h1' and b.tok16 and not a.tok16          # b: the token-major fp16 pipeline (csrc/tdr_tok16.hip)
        from textualdegremoval_amd import dino as _dino
        _dino.TOK16 = False
        try:
            c = DinoMatcher(sdb, torch.device('cuda'), linear_math='h1')    # the same arithmetic on the channel-major engines
        finally:
            _dino.TOK16 = True
        xs = images(3, 126, 126, seed=77).cuda()
        (fb, Tb), (fc, Tc), (fa, _) = b.tokens(xs, flat=True), c.tokens(xs, flat=True), a.tokens(xs, flat=True)
        assert Tb == Tc and not c.tok16
        e_bc, e_ca = (fb - fc).view(3, 768, -1)[..., :Tb + 1].abs().max().item(), (fc - fa).view(3, 768, -1)[..., :Tb + 1].abs().max().item()
        print(f'final-norm tokens: token-major fp16 pipeline vs channel-major h1 {e_bc:.2e}; channel-major h1 vs the 2-way split {e_ca:.2e}')
        assert e_bc < 2 * e_ca + 1e-4                                        # two orderings of one arithmetic: inside its own rounding noise
        worst_margin, n = 1.0, 0
        for seed in range(4):
            big = images(2, 256, 256, seed=40 + seed)
            for oy, ox in ((0, 0), (32, 96), (128, 64), (96, 128)):
                lqc = (big[:, :, oy:oy + 128, ox:ox + 128] + 15 / 255 * torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(seed))).contiguous()
                ra, ia, ca = a.match(lqc.cuda(), big.cuda())
                rb, ib, cb = b.match(lqc.cuda(), big.cuda())
                assert torch.equal(ia, ib) and torch.equal(ra, rb), (seed, oy, ox)
                top2 = torch.topk(ca, 2, dim=1).values
                worst_margin = min(worst_margin, float((top2[:, 0] - top2[:, 1]).min()))
                assert (ca - cb).abs().max().item() < 0.25 * float((top2[:, 0] - top2[:, 1]).min()) + 1e-6
                n += 2
        print(f'single-product DINO linears: {n} matches over 25 windows each identical to the split arithmetic; smallest top-1 / top-2 '
              f'similarity margin {worst_margin:.4f}')
    finally:
        K.set_math(prev)


def test_tok16x3_kernels_against_torch():
    """csrc/tdr_tok16.hip on 3-way bf16 split planes (the default 'bx3' arithmetic): LayerNorm -> planes, the GEMM with its three
    epilogues (+ LayerScale), channel-major -> planes, against fp64 torch on the fp32 values.  h + m + l IS the fp32 value, so the
    planes are compared exactly; the GEMM bar is a few 2^-24 of the accumulated magnitude (sum |x||w|) -- fp32-class, not bf16."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    g = torch.Generator().manual_seed(8)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    md = lambda a, b: (a.double() - b.double()).abs().max().item()
    sum3 = lambda p: (p[0].float() + p[1].float()) + p[2].float()
    P, D = 2 * 128 + 24, 768                                             # a ragged last row tile
    t, w, b = r(P, D) * 2 + 0.5, r(D), r(D)
    pl = K.tok_layernorm(t, w, b, 1e-6, planes=3)
    y32 = K.tok_layernorm(t, w, b, 1e-6, out_f16=False)
    assert pl.shape == (3, P, D) and pl.dtype == torch.bfloat16 and torch.equal(sum3(pl), y32)      # the split loses nothing
    assert md(y32, torch.nn.functional.layer_norm(t.double(), (D,), w.double(), b.double(), 1e-6)) < 2e-6 * y32.abs().max().item()
    a = r(96, 72) * torch.logspace(-30, 20, 72).cuda()                     # any fp32 exponent: no fp16 window
    ap = K.cm_to_tok16x3(a)
    assert torch.equal(sum3(ap), a.t().contiguous())
    x = r(5, 7)
    assert torch.equal(sum3(K.split_planes3(x)), x)
    for Pn in (P, 4 * 1024 + 8):                                         # the 64-row tiles of a short launch / the 128-row tiles of a wide one
        for N, Kd in ((384, 768), (768, 160)):
            x, wt, bias, ls = r(Pn, Kd), r(N, Kd) * 0.05, r(N), r(N)
            x3, w3 = K.split_planes3(x), K.split_planes3(wt)
            acc = x.double() @ wt.double().t() + bias.double()
            bar = 8 * 2 ** -24 * (x.abs().double() @ wt.abs().double().t()).max().item()
            assert md(K.tok16x3_gemm(x3, w3, bias, epi=3), acc.t()) < bar
            res = r(Pn, N)
            assert md(K.tok16x3_gemm(x3, w3, bias, epi=2, out32=res.clone()), res.double() + acc) < bar + 1e-6
            assert md(K.tok16x3_gemm(x3, w3, bias, epi=2, out32=res.clone(), ls=ls), res.double() + ls.double() * acc) < 4 * bar + 1e-6
            for act, f in ((0, lambda v: v), (2, torch.nn.functional.gelu), (3, lambda v: v * torch.sigmoid(1.702 * v))):
                y = K.tok16x3_gemm(x3, w3, bias, epi=4, act=act)
                assert y.dtype == torch.bfloat16 and md(sum3(y), f(acc)) < 2 * bar + 2e-7 * acc.abs().max().item(), act
            assert md(K.tok16x3_gemm(x3, w3, None, epi=3), (acc - bias.double()).t()) < bar
    # operands far outside the fp16 range: the 3-way split has no window
    x, wt = r(136, 64) * 1e6, r(128, 64) * 1e-9
    acc = x.double() @ wt.double().t()
    assert md(K.tok16x3_gemm(K.split_planes3(x), K.split_planes3(wt), None, epi=3), acc.t()) < 8 * 2 ** -24 * (x.abs().double() @ wt.abs().double().t()).max().item()


def test_default_matcher_on_the_token_major_triple_planes():
    """DinoMatcher at the library default (bx3) and ViT-B/14 geometry with dino.TOK16X3 = True: flat passes run the blocks on tdr_tok16x3_gemm (operands pre-split
    into h | m | l bf16 planes, token-major).  One arithmetic, two layouts: tokens against the channel-major engines (the default)
    and against the oracle at fp32-class bars, every window decision identical, and identical to the exact-fp32 matcher's."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.dino import DinoMatcher
    prev = K.MATH
    K.set_math('bx3')
    try:
        sd = D.synth_vit_params(768, 2, 12, seed=5)
        from textualdegremoval_amd import dino as _dino
        _dino.TOK16X3 = True                                               # opt-in path (measured neutral on the matcher-active step)
        try:
            m = DinoMatcher(sd, torch.device('cuda'), heads=12)
        finally:
            _dino.TOK16X3 = False
        assert m.tok16x3 and not m.tok16 and m.linear_math is None
        c = DinoMatcher(sd, torch.device('cuda'), heads=12)
        assert not c.tok16x3
        xs = images(3, 126, 98, seed=77)
        (fm, Tm), (fc, Tc) = m.tokens(xs.cuda(), flat=True), c.tokens(xs.cuda(), flat=True)
        ref = D.vit_patch_tokens(sd, xs, heads=12)                                  # [B, T, D]
        tm = fm.reshape(3, 768, -1)[:, :, 1:Tm + 1].transpose(1, 2).cpu()
        tc = fc.reshape(3, 768, -1)[:, :, 1:Tc + 1].transpose(1, 2).cpu()
        e_mc, e_mo, e_co = (tm - tc).abs().max().item(), (tm - ref).abs().max().item(), (tc - ref).abs().max().item()
        print(f'final-norm tokens: token-major planes vs channel-major {e_mc:.2e}; vs the oracle {e_mo:.2e} (channel-major vs the oracle {e_co:.2e})')
        assert Tm == Tc and e_mc < 2e-5 * ref.abs().max().item() and e_mo < 1e-4
        K.set_math('f32')
        f = DinoMatcher(sd, torch.device('cuda'), heads=12, linear_math='f32')
        K.set_math('bx3')
        for seed in range(2):
            big = images(2, 256, 256, seed=40 + seed)
            for oy, ox in ((0, 0), (32, 96), (128, 64)):
                lqc = (big[:, :, oy:oy + 128, ox:ox + 128] + 15 / 255 * torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(seed))).contiguous()
                rm, im, cm = m.match(lqc.cuda(), big.cuda())
                rc, ic, cc = c.match(lqc.cuda(), big.cuda())
                K.set_math('f32')
                rf, i_f, cf = f.match(lqc.cuda(), big.cuda())
                K.set_math('bx3')
                assert torch.equal(im, ic) and torch.equal(im, i_f) and torch.equal(rm, rc), (seed, oy, ox)
                assert (cm - cf).abs().max().item() < 1e-4 and (cm - cc).abs().max().item() < 1e-5
    finally:
        K.set_math(prev)


@pytest.mark.parametrize('shape', [(3, 12, 64, 300), (2, 16, 80, 257), (2, 2, 32, 97), (1, 4, 16, 64)])
def test_attention_on_the_bf16_split_against_float64(shape):
    """tdr_attention_fwd_math code 1 (round 6: the frozen ViTs' attention in the default arithmetic -- q, k, v and P as three bf16 planes,
    six products, fp32 softmax) against a float64 reference and beside the exact fp32 MFMA kernel it replaces there: at least as close to
    float64 as the exact kernel (measured 8.4e-6 against 1.4e-5 at the matcher's shape), padding columns zero, ragged token counts"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import _lib, kernels as K
    lib = _lib.load()
    B, heads, hd, T = shape
    LD = (T + 31) // 32 * 32
    g = torch.Generator().manual_seed(hd + T)
    qkv = (torch.randn(B, 3 * heads * hd, LD, generator=g) * 1.5).cuda()
    scale = hd ** -0.5
    outs = {}
    for math in (0, 1):
        out = torch.full((B, heads * hd, LD), 7.0, device='cuda')
        _lib.check(lib.tdr_attention_fwd_math(qkv.data_ptr(), B, heads * hd, heads, T, LD, scale, math, 0, out.data_ptr(), K._stream()), 'attn')
        outs[math] = out.cpu().double()
    q, k, v = (t.double().cpu().view(B, heads, hd, LD)[..., :T] for t in qkv.chunk(3, 1))
    ref = torch.einsum('bhqk,bhdk->bhdq', torch.softmax(torch.einsum('bhdq,bhdk->bhqk', q, k) * scale, -1), v).reshape(B, heads * hd, T)
    e0, e1 = (outs[0][..., :T] - ref).abs().max().item(), (outs[1][..., :T] - ref).abs().max().item()
    assert e1 < 2e-5 and e1 <= 1.5 * e0 + 2e-6, (e0, e1)
    assert float(outs[1][..., T:].abs().max()) == 0.0 if LD > T else True

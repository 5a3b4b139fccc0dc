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
    yield DinoMatcher(sd, torch.device('cuda'), heads=2, linear_math=request.param), sd      # (the default would be 'h1' under hx2)
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


def test_single_product_linears_keep_every_match_decision():
    """DinoMatcher(linear_math='h1'): the frozen Linears on ONE fp16 MFMA product.  The sub-graph's only output is the window index, so
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
        sd = D.synth_vit_params(32, 2, 2, seed=11)
        m1 = DinoMatcher(sd, torch.device('cuda'), heads=2, linear_math='h1')
        gold = np.load(os.path.join(GOLDEN, 'dino_vit_e32_d2.npz'))
        lq, ref = match_inputs()
        ref_in, idx, corr = m1.match(lq.cuda(), ref.cuda())
        o_ref_in, o_idx, o_corr = D.match_reference_window(sd, lq, ref, heads=2)
        assert np.array_equal(idx.cpu().numpy(), gold['match_index']) and torch.equal(ref_in.cpu(), o_ref_in)
        assert (corr.cpu() - o_corr).abs().max().item() < 5e-3
        sdb = random_vit_b14_state_dict(seed=1, depth=12)
        a, b = DinoMatcher(sdb, torch.device('cuda'), linear_math='hx2'), DinoMatcher(sdb, torch.device('cuda'))     # default: 'h1'
        assert b.linear_math == 'h1'
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

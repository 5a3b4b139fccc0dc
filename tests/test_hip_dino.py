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
    yield DinoMatcher(sd, torch.device('cuda'), heads=2), sd
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
    m = DinoMatcher(sd, torch.device('cuda'), heads=12)
    x = images(1, 70, 70, seed=9)
    tok, T = m.tokens(x.cuda())
    t = tok.reshape(1, 768, -1)[:, :, 1:T + 1].transpose(1, 2).cpu()
    ref = D.vit_patch_tokens(sd, x, heads=12)
    assert (t - ref).abs().max().item() < 2e-4

"""CPU: the DINOv2 oracle restatement against the golden vectors recorded from the reference classes
(tests/golden/make_golden_dino.py)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import dino_oracle as D

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def images(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(H // 16, 2), max(W // 16, 2), generator=g)
    return F.interpolate(low, size=(H, W), mode='bicubic').clamp(0, 1)


def match_inputs():
    B, h = 2, 48
    clean = images(B, 96, 96, seed=7)
    g = torch.Generator().manual_seed(8)
    lq = clean[:, :, 24:72, 12:60] + torch.randn(B, 3, h, h, generator=g) * (15 / 255)
    return lq, clean


def test_patch_tokens_match_reference():
    gold = np.load(os.path.join(GOLDEN, 'dino_vit_e32_d2.npz'))
    sd = D.synth_vit_params(32, 2, 2, seed=11)
    for tag, (B, H, W) in {'sq56': (2, 56, 56), 'rect70x42': (1, 70, 42), 'sq140': (1, 140, 140)}.items():
        y = D.vit_patch_tokens(sd, images(B, H, W, seed=100 + H), heads=2)
        assert np.abs(y.numpy() - gold[f'tokens_{tag}']).max() < 2e-5, tag


def test_window_match_matches_reference():
    gold = np.load(os.path.join(GOLDEN, 'dino_vit_e32_d2.npz'))
    sd = D.synth_vit_params(32, 2, 2, seed=11)
    lq, ref = match_inputs()
    ref_in, idx, corr = D.match_reference_window(sd, lq, ref, heads=2)
    assert np.array_equal(idx.numpy(), gold['match_index'])
    assert np.abs(corr.numpy() - gold['match_corr']).max() < 1e-5
    s = np.array([ref_in.double().sum().item(), ref_in.double().abs().sum().item()])
    assert np.allclose(s, gold['match_ref_in_sum'], rtol=0, atol=1e-6)
    assert gold['match_gap'].min() > 1e-3           # the fixture is not a near-tie

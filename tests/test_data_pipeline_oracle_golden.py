"""oracle/data_pipeline_oracle.py against tests/golden/transforms.npz -- vectors produced by running the reference's own
data/transforms.py (paired_random_crop, data_augmentation, random_augmentation) and the dataset's noise statements
(restoration_dataset.py:465-476), see tests/golden/make_golden_transforms.py.  CPU only."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data_pipeline_oracle as DO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transforms.npz'))


def image(seed, h, w):
    return np.random.RandomState(seed).rand(h, w, 3).astype(np.float32)


@pytest.mark.parametrize('mode', range(8))
def test_augment_modes(mode):
    assert np.array_equal(DO.augment_mode(image(1, 5, 7), mode), G[f'mode{mode}'])


@pytest.mark.parametrize('case', [tuple(int(v) for v in c) for c in G['cases']])
def test_crop_then_augment_same_draws_same_pixels(case):
    seed, h, w, patch = case
    rng = random.Random(seed)
    g, q, draws = DO.train_sample(image(seed, h, w), image(seed + 1000, h, w), patch, rng)
    assert list(draws) == [int(v) for v in G[f'c{seed}_draws']]
    assert np.array_equal(g, G[f'c{seed}_gt']) and np.array_equal(q, G[f'c{seed}_lq'])


def test_small_image_without_padding_raises_like_the_reference():
    assert int(G['small_raises']) == 1
    with pytest.raises(ValueError):
        DO.train_sample(image(2, 20, 20), image(3, 20, 20), 32, random.Random(0), pad=False)
    g, q, _ = DO.train_sample(image(2, 20, 20), image(3, 20, 20), 32, random.Random(0))      # with padding(): fine
    assert g.shape == (32, 32, 3)


@pytest.mark.parametrize('tag,stype,srange', [('const', 'constant', 15), ('rand', 'random', [0, 55]), ('choice', 'choice', [15, 25, 50])])
def test_sigma_noise(tag, stype, srange):
    img = image(30, 24, 20).transpose(2, 0, 1).copy()
    rng = random.Random(77)
    sigma = DO.draw_sigma(stype, srange, rng)
    assert float(sigma) == float(G[f'noise_{tag}_sigma'])
    torch.manual_seed(78)
    randn = torch.randn(3, 24, 20).numpy()
    out = DO.add_noise(img, sigma, randn)
    assert np.abs(out - G[f'noise_{tag}_out']).max() <= 6e-8          # one fp32 rounding of the product


def test_padding_is_bottom_right_reflection_including_the_edge():
    a = np.arange(12, dtype=np.float32).reshape(3, 4, 1)
    p = DO.padding(a, 6)
    assert p.shape == (6, 6, 1)
    assert np.array_equal(p[:3, :4], a) and np.array_equal(p[3], p[2]) and np.array_equal(p[5], p[0])
    assert np.array_equal(p[:, 4], p[:, 3]) and np.array_equal(p[:, 5], p[:, 2])

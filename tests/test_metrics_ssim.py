"""SSIM / Y-channel metrics (metrics/psnr_ssim.py:131-300 of the reference).  CPU: the oracle's own consistency and the host
colour conversion; GPU: csrc/tdr_metrics.hip through textualdegremoval_amd.metrics.calculate_ssim against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import metrics_oracle as MO  # noqa: E402
from textualdegremoval_amd import metrics as M  # noqa: E402


def _pair(h, w, c, seed, uint8=True, noise=12.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(yy / 7.0)[..., None] * np.cos(xx / 5.0)[..., None] + rng.normal(0, 20, (h, w, c))
    a = np.clip(base, 0, 255)
    b = np.clip(base + rng.normal(0, noise, (h, w, c)), 0, 255)
    if uint8:
        return a.round().astype(np.uint8), b.round().astype(np.uint8)
    return (a / 255).astype(np.float32), (b / 255).astype(np.float32)


def test_oracle_window_and_identity():
    g = MO.gaussian_kernel_11()
    assert abs(g.sum() - 1) < 1e-15 and np.allclose(g, g[::-1]) and g.argmax() == 5
    a, _ = _pair(24, 31, 3, 0)
    assert abs(MO.calculate_ssim(a, a, 0) - 1.0) < 1e-6
    assert abs(MO.calculate_ssim(a, a, 0, test_y_channel=True) - 1.0) < 1e-12


def test_oracle_single_channel_volume_equals_2d_replicate_window():
    # with C = 1 the replicate-padded channel axis sums the window to one: _ssim_3d degenerates to _ssim_cly's filtering
    a, b = _pair(33, 29, 1, 1)
    v3 = MO.ssim_3d(a.astype(np.float64), b.astype(np.float64), 255)
    v2 = MO.ssim_cly(a[..., 0], b[..., 0])
    assert abs(v3 - v2) < 2e-5


def test_host_y_channel_psnr_matches_oracle():
    a, b = _pair(40, 36, 3, 2)
    got = M.calculate_psnr(a, b, crop_border=2, test_y_channel=True)
    assert abs(got - MO.calculate_psnr_y(a, b, 2)) < 1e-9
    ya = M.to_y_channel(a.astype(np.float64))
    assert ya.shape == (40, 36, 1) and 16.0 <= ya.min() and ya.max() <= 235.0 + 1e-3


def test_ssim_needs_the_device():
    a, b = _pair(16, 16, 3, 3)
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(Exception):
        M.calculate_ssim(a, b, 0)


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,c,uint8,crop', [(64, 64, 3, True, 0), (37, 53, 3, True, 4), (48, 40, 1, True, 0),
                                              (50, 70, 3, False, 0), (16, 16, 3, True, 0), (7, 9, 3, True, 0),
                                              (96, 80, 4, False, 3), (33, 17, 2, True, 0)])
def test_hip_ssim_matches_oracle(h, w, c, uint8, crop):
    a, b = _pair(h, w, c, 10 + h + w, uint8)
    want = MO.calculate_ssim(a, b, crop)
    got = M.calculate_ssim(a, b, crop)
    # float32 like the reference (img.float().cuda()): sigma = E[x^2] - mu^2 cancels at the 255^2 magnitude, so the summation
    # order (dense 11^3 conv there, separable here) shows at 1e-5 on [0,255] images; [0,1] images agree to 2e-5
    tol = 1e-4 if uint8 else 2e-5
    assert abs(got - want) < tol, (got, want)
    assert abs(M.calculate_ssim(a, a, crop) - 1.0) < 1e-5


@pytest.mark.gpu
def test_hip_ssim_input_orders_and_y_channel():
    a, b = _pair(45, 52, 3, 77)
    want = MO.calculate_ssim(a, b, 0)
    chw = M.calculate_ssim(np.ascontiguousarray(a.transpose(2, 0, 1)), np.ascontiguousarray(b.transpose(2, 0, 1)), 0, input_order='CHW')
    assert abs(chw - want) < 1e-4
    ta = torch.from_numpy(a.transpose(2, 0, 1).astype(np.float32))[None]
    tb = torch.from_numpy(b.transpose(2, 0, 1).astype(np.float32))[None]
    assert abs(M.calculate_ssim(ta, tb, 0) - want) < 1e-4
    wy = MO.calculate_ssim(a, b, 3, test_y_channel=True)
    gy = M.calculate_ssim(a, b, 3, test_y_channel=True)
    assert abs(gy - wy) < 1e-4, (gy, wy)
    with pytest.raises(ValueError):
        M.calculate_ssim(a, b, 0, input_order='WHC')


@pytest.mark.gpu
def test_hip_ssim_full_size_properties():
    # validation-image size: symmetry, identity, monotone in the noise level
    a, b1 = _pair(512, 768, 3, 5, noise=5.0)
    _, b2 = _pair(512, 768, 3, 5, noise=25.0)
    s1, s2 = M.calculate_ssim(a, b1, 0), M.calculate_ssim(a, b2, 0)
    assert 0 < s2 < s1 < 1
    assert abs(M.calculate_ssim(b1, a, 0) - s1) < 1e-6
    assert abs(M.calculate_ssim(a, a, 0) - 1.0) < 1e-5

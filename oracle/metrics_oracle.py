"""CPU restatement of the reference's SSIM (metrics/psnr_ssim.py) -- TEST INFRASTRUCTURE ONLY: imported by tests/, never by
the product (textualdegremoval_amd.metrics.calculate_ssim runs csrc/tdr_metrics.hip and has no host fallback).

PINNED (round 4) for the Y-channel path: `ssim_cly` is checked against `_ssim_cly` executed from the reference file
(tests/golden/make_golden_tlsc.py: the two cv2 calls it makes are served by scipy.ndimage.correlate(mode='nearest') and the
kernel's published formula), tests/test_hip_tlsc.py.  PARITY UNPINNED for `ssim_3d`: the reference moves its conv3d to
.cuda() and imports cv2 (absent from this image), so that function cannot be run here to make golden vectors.  cv2.getGaussianKernel(11, 1.5) is restated from its published definition
(G_i = alpha * exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)), sum 1; ksize 11 > 7 so no fixed table applies), and the
conv3d / filter2D calls are restated with torch / numpy.
"""
import numpy as np
import torch
import torch.nn.functional as F


def gaussian_kernel_11():
    x = np.arange(11, dtype=np.float64) - 5.0
    g = np.exp(-(x * x) / (2.0 * 1.5 * 1.5))
    return g / g.sum()


def ssim_3d(img1, img2, max_value):
    """_ssim_3d (:131-176): float32 conv3d of the [H,W,C] volume with the dense 11^3 window, replicate padding 5"""
    g = gaussian_kernel_11()
    window = np.outer(g, g)
    kern = torch.tensor(np.stack([window * k for k in g], axis=0)).float()[None, None]
    c1, c2 = (0.01 * max_value) ** 2, (0.03 * max_value) ** 2
    a = torch.tensor(img1.astype(np.float64)).float()
    b = torch.tensor(img2.astype(np.float64)).float()

    def filt(v):
        v = F.pad(v[None, None], (5, 5, 5, 5, 5, 5), mode='replicate')
        return F.conv3d(v, kern)[0, 0]
    mu1, mu2 = filt(a), filt(b)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = filt(a ** 2) - mu1_sq
    s2 = filt(b ** 2) - mu2_sq
    s12 = filt(a * b) - mu1_mu2
    m = ((2 * mu1_mu2 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return float(m.mean())


def ssim_cly(img1, img2):
    """_ssim_cly (:184-222): float64, 2-D 11x11 window, BORDER_REPLICATE, [0,255] constants"""
    g = gaussian_kernel_11()
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)

    def filt(v):
        p = np.pad(v, 5, mode='edge')
        t = sum(g[k] * p[:, k:k + v.shape[1]] for k in range(11))
        return sum(g[k] * t[k:k + v.shape[0], :] for k in range(11))
    mu1, mu2 = filt(a), filt(b)
    s1 = filt(a * a) - mu1 ** 2
    s2 = filt(b * b) - mu2 ** 2
    s12 = filt(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))
    return float(m.mean())


def to_y_channel(img):
    """metrics/metric_util.py:34-47 + utils/matlab_functions.py:207-238 (y_only)"""
    img = img.astype(np.float32) / 255.
    if img.ndim == 3 and img.shape[2] == 3:
        img = ((np.dot(img, [24.966, 128.553, 65.481]) + 16.0) / 255.).astype(np.float32)[..., None]
    return img * 255.


def calculate_ssim(img1, img2, crop_border, test_y_channel=False):
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    if a.ndim == 2:
        a, b = a[..., None], b[..., None]
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel:
        a, b = to_y_channel(a), to_y_channel(b)
        return ssim_cly(a[..., 0], b[..., 0])
    return ssim_3d(a, b, 1 if a.max() <= 1 else 255)


def calculate_psnr_y(img1, img2, crop_border):
    """calculate_psnr with test_y_channel=True (:9-63)"""
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    a, b = to_y_channel(a), to_y_channel(b)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20. * np.log10((1. if a.max() <= 1 else 255.) / np.sqrt(mse))

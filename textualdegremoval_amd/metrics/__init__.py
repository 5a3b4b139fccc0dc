"""PSNR / SSIM with the reference's semantics (metrics/psnr_ssim.py, utils/utils_image.py:129-192): float64 MSE on
[0,255] uint8 images or [0,1] floats; SSIM with the 11^3 Gaussian over the [H,W,C] volume on the device
(csrc/tdr_metrics.hip -- the reference runs its conv3d on the GPU too, psnr_ssim.py:152-156), or on the Y channel with the
2-D replicate-border window (:184-222)."""
import numpy as np
import torch


def tensor2img(tensor, rgb2bgr=True, out_type=np.uint8, min_max=(0, 1)):
    t = tensor.squeeze(0).float().detach().cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    img = t.numpy()
    if img.ndim == 3:
        img = img.transpose(1, 2, 0)
        if img.shape[2] == 1:
            img = img[:, :, 0]
        elif rgb2bgr:
            img = img[:, :, ::-1]
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return np.ascontiguousarray(img.astype(out_type))


def calculate_psnr(img1, img2, crop_border=0, input_order='HWC', test_y_channel=False):
    if input_order not in ('HWC', 'CHW'):
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')

    def to_np(x):
        if isinstance(x, torch.Tensor):
            if x.dim() == 4:
                x = x.squeeze(0)
            x = x.detach().cpu().numpy().transpose(1, 2, 0)
        elif input_order == 'CHW':
            x = x.transpose(1, 2, 0)
        return x.astype(np.float64)
    a, b = to_np(img1), to_np(img2)
    assert a.shape == b.shape, f'Image shapes are differnet: {a.shape}, {b.shape}.'
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border, ...]
        b = b[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        a, b = to_y_channel(a), to_y_channel(b)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    max_value = 1. if a.max() <= 1 else 255.
    return 20. * np.log10(max_value / np.sqrt(mse))


def bgr2ycbcr_y(img):
    """Y of ITU-R BT.601 YCbCr from a BGR float32 image in [0,1] (utils/matlab_functions.py:207-238, y_only branch)"""
    # np.dot with the python-float coefficients is float64; the reference divides by 255 in float64 and casts afterwards
    # (_convert_output_type_range, utils/matlab_functions.py:354-361): casting first is off by one float32 ulp in some pixels
    return ((np.dot(img.astype(np.float32), [24.966, 128.553, 65.481]) + 16.0) / 255.).astype(np.float32)


def to_y_channel(img):
    """[0,255] image -> its Y channel, [0,255] float, unrounded (metrics/metric_util.py:34-47)"""
    img = img.astype(np.float32) / 255.
    if img.ndim == 3 and img.shape[2] == 3:
        img = bgr2ycbcr_y(img)[..., None]
    return img * 255.


def _hwc(x, input_order):
    if isinstance(x, torch.Tensor):
        if x.dim() == 4:
            x = x.squeeze(0)
        x = x.detach().cpu().numpy().transpose(1, 2, 0)
    if x.ndim == 2:
        x = x[..., None]
    elif input_order == 'CHW':
        x = x.transpose(1, 2, 0)
    return x.astype(np.float64)


def calculate_ssim(img1, img2, crop_border, input_order='HWC', test_y_channel=False):
    """metrics/psnr_ssim.py:224-300.  The filtering runs on the device (K.ssim3d); there is no host fallback."""
    from .. import kernels as K
    assert img1.shape == img2.shape, f'Image shapes are differnet: {img1.shape}, {img2.shape}.'
    if input_order not in ('HWC', 'CHW'):
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    a, b = _hwc(img1, input_order), _hwc(img2, input_order)
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border, ...]
        b = b[crop_border:-crop_border, crop_border:-crop_border, ...]
    dev = torch.device('cuda', torch.cuda.current_device())
    if test_y_channel:
        # _ssim_cly (:184-222): float64 filtering of the Y planes (csrc/tdr_tlsc.hip, ssim_y64_kernel: every field in double)
        a, b = to_y_channel(a), to_y_channel(b)
        ya = torch.from_numpy(np.ascontiguousarray(a[..., 0], dtype=np.float32)).to(dev)
        yb = torch.from_numpy(np.ascontiguousarray(b[..., 0], dtype=np.float32)).to(dev)
        return K.ssim_y64(ya, yb)
    max_value = 1 if a.max() <= 1 else 255
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(dev)
    return float(K.ssim3d(ta, tb, max_value).item())

"""PSNR with the reference's semantics (metrics/psnr_ssim.py:9-63,
utils/utils_image.py:129-192): float64 MSE on [0,255] uint8 images or [0,1] floats."""
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
    if test_y_channel:
        raise NotImplementedError('Y-channel PSNR is not on the restoration train path')

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
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    max_value = 1. if a.max() <= 1 else 255.
    return 20. * np.log10(max_value / np.sqrt(mse))

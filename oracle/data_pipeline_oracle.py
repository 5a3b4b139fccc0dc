"""CPU ORACLE (test infrastructure, NOT product code).

numpy restatement of the reference's host-side input pipeline for one sample: paired_random_crop
(data/transforms.py:24-84) with the crop origin given, data_augmentation modes 0-7 (:223-270: compositions of np.flipud and
np.rot90 on an HWC image) and the sigma-noise synthesis lq = gt + noise * sigma / 255 (data/restoration_dataset.py:464-476).
"parity unpinned": data/transforms.py imports cv2, which this image lacks, so the reference module cannot be executed here
to generate vectors; the functions below are the numpy calls the reference makes, written out.

Only tests/ may import this module."""
import numpy as np


def augment_mode(img_hwc, mode):
    if mode == 0:
        return img_hwc
    if mode == 1:
        return np.flipud(img_hwc)
    if mode == 2:
        return np.rot90(img_hwc)
    if mode == 3:
        return np.flipud(np.rot90(img_hwc))
    if mode == 4:
        return np.rot90(img_hwc, k=2)
    if mode == 5:
        return np.flipud(np.rot90(img_hwc, k=2))
    if mode == 6:
        return np.rot90(img_hwc, k=3)
    if mode == 7:
        return np.flipud(np.rot90(img_hwc, k=3))
    raise Exception('Invalid choice of image transformation')


def crop_augment(img_chw, top, left, patch, mode, noise_chw=None, sigma=None):
    hwc = np.transpose(img_chw, (1, 2, 0))[top:top + patch, left:left + patch, :]
    out = np.ascontiguousarray(np.transpose(augment_mode(hwc, mode), (2, 0, 1)))
    if noise_chw is not None:
        out = out + noise_chw * (1.0 if sigma is None else sigma)
    return out

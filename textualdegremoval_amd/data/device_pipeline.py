"""Input pipeline on the device.  The reference augments on the host, per sample, inside its Dataset
`__getitem__` (numpy slicing, np.flipud / np.rot90, torch.randn on the CPU): data/transforms.py:24-84 (paired_random_crop),
:223-275 (data_augmentation / random_augmentation), data/restoration_dataset.py:464-476 (sigma-noise synthesis).  On an
MI355X the batch is already resident in HBM, so the same three steps are one gather kernel over the whole batch
(`tdr_crop_augment`); only the few random PARAMETERS are drawn on the host, with python's `random` module in the same order
per sample as the reference (top, left, then the augmentation flag, then sigma), so a seeded run picks the same crops.
Images smaller than gt_size are reflect-padded at the bottom / right first, as the reference's padding() does
(utils/utils_image.py:243-259; folded into the gather).  The reference image `ref` is returned UNTOUCHED: the WithRef
datasets crop and augment only (img_gt, img_lq) -- `random_augmentation(img_gt, img_lq)`, restoration_dataset.py:140,235,
459,601,756 -- and hand img_ref through as loaded; the model matches / crops it itself with DINOv2."""
import random

import torch

from .. import kernels as K


class DevicePairedAugmenter:
    """gt / lq / ref full images [N,3,H,W] on the device -> training patches.

    opt keys follow the reference's dataset options: gt_size, scale (1 on this path), geometric_augs (bool), sigma_type
    ('constant' | 'random' | 'choice' | None) and sigma_range as in restoration_dataset.py:378-380,464-470."""

    def __init__(self, opt, rng=None):
        self.patch = int(opt['gt_size'])
        self.scale = int(opt.get('scale', 1))
        if self.scale != 1:
            raise NotImplementedError('device pipeline: scale 1 (restoration, not super-resolution)')
        self.geometric_augs = bool(opt.get('geometric_augs', True))
        self.sigma_type = opt.get('sigma_type')
        self.sigma_range = opt.get('sigma_range')
        self.rng = rng or random

    def draw(self, N, H, W):
        """host side: the per-sample parameters, drawn as the reference draws them (random.randint is inclusive)."""
        top, left, mode, sigma = [], [], [], []
        H, W = max(H, self.patch), max(W, self.patch)     # after the reference's padding() to at least gt_size
        for _ in range(N):
            top.append(self.rng.randint(0, H - self.patch))
            left.append(self.rng.randint(0, W - self.patch))
            mode.append(self.rng.randint(0, 7) if self.geometric_augs else 0)
            if self.sigma_type == 'constant':
                sigma.append(float(self.sigma_range))
            elif self.sigma_type == 'random':
                sigma.append(self.rng.uniform(self.sigma_range[0], self.sigma_range[1]))
            elif self.sigma_type == 'choice':
                sigma.append(float(self.rng.choice(self.sigma_range)))
        return top, left, mode, sigma

    def __call__(self, gt, lq=None, ref=None, generator=None):
        """returns dict(lq, gt[, ref]) of patches.  lq None + sigma_type set: lq = gt patch + N(0, (sigma/255)^2)."""
        N, _, H, W = gt.shape
        top, left, mode, sigma = self.draw(N, H, W)
        dev = gt.device
        t = torch.tensor(top, dtype=torch.int32, device=dev)
        l = torch.tensor(left, dtype=torch.int32, device=dev)
        m = torch.tensor(mode, dtype=torch.int32, device=dev)
        out = {'gt': K.crop_augment(gt, t, l, m, self.patch)}
        if lq is not None:
            out['lq'] = K.crop_augment(lq, t, l, m, self.patch)
        elif sigma:
            noise = torch.randn(N, gt.shape[1], self.patch, self.patch, device=dev, generator=generator)
            s = torch.tensor(sigma, dtype=torch.float32, device=dev) / 255.0
            out['lq'] = K.crop_augment(gt, t, l, m, self.patch, noise=noise, sigma=s)
        if ref is not None:
            out['ref'] = ref          # never cropped, never augmented by the reference's datasets (module docstring)
        self.last = dict(top=top, left=left, mode=mode, sigma=sigma)
        return out

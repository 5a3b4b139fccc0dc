"""Input pipeline on the device.  The reference augments on the host, per sample, inside its Dataset
`__getitem__` (numpy slicing, np.flipud / np.rot90, torch.randn on the CPU): data/transforms.py:24-84 (paired_random_crop),
:223-275 (data_augmentation / random_augmentation), data/restoration_dataset.py:464-476 (sigma-noise synthesis).  On an
MI355X the batch is already resident in HBM, so the same three steps are one gather kernel over the whole batch
(`tdr_crop_augment`); only the few random PARAMETERS are drawn on the host, with python's `random` module in the same order
per sample as the reference (top, left, then the augmentation flag, then sigma), so a seeded run picks the same crops."""
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
            # the reference image goes through the same geometric mode (random_augmentation(*args) applies one flag to all
            # its arguments, transforms.py:271-275); it is not cropped here -- the model matches / crops it itself
            out['ref'] = K.crop_augment(ref, None, None, m, ref.shape[-1]) if ref.shape[-1] == ref.shape[-2] else ref
        self.last = dict(top=top, left=left, mode=mode, sigma=sigma)
        return out

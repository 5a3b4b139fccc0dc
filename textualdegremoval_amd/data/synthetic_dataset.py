"""Synthetic paired-with-reference dataset (SURVEY 8d recipe): gt = clamp(bicubic-up(U[0,1] at 1/32 res)),
lq = gt + N(0,(sigma/255)^2) (the sigma-noise synthesis of data/restoration_dataset.py:465-476), ref = the clean
image at `ref_size` (>= gt_size: the model crops / matches it itself, image_restoration_ref_model.py:219-243).
Sample i is a pure function of (seed, i).  Returned keys follow Dataset_*WithRef: lq, gt, ref, lq_path, gt_path."""
import torch
import torch.nn.functional as F
from torch.utils import data


class Dataset_SyntheticPairedWithRef(data.Dataset):
    def __init__(self, opt):
        self.opt = opt
        self.n = int(opt.get('num_images', 64))
        self.size = int(opt.get('gt_size', 128))
        self.ref_size = int(opt.get('ref_size', self.size))
        self.sigma = float(opt.get('sigma', 15.0))
        self.seed = int(opt.get('seed', 1234))
        if self.ref_size < self.size:
            raise ValueError('ref_size must be >= gt_size')

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed + int(index))
        big = self.ref_size
        low = torch.rand(1, 3, max(big // 32, 2), max(big // 32, 2), generator=g)
        clean = F.interpolate(low, size=(big, big), mode='bicubic', align_corners=False).clamp(0, 1)[0]
        o = (big - self.size) // 2 // max(self.size // 4, 1) * max(self.size // 4, 1)
        gt = clean[:, o:o + self.size, o:o + self.size].contiguous()
        lq = gt + torch.randn(3, self.size, self.size, generator=g) * (self.sigma / 255.0)
        return {'lq': lq, 'gt': gt, 'ref': clean, 'lq_path': f'synthetic/{index:06d}', 'gt_path': f'synthetic/{index:06d}'}

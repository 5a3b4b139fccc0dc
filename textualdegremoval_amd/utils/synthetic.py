"""Synthetic restoration pairs: gt = clamp(bicubic-up(U[0,1] at 1/32 res)),
ref = gt, lq = gt + N(0,(sigma/255)^2) -- the sigma-noise recipe of the reference's
denoise dataset (data/restoration_dataset.py:465-476)."""
import torch
import torch.nn.functional as F


def synthetic_pair(B, H, W, seed=1234, sigma=15.0):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(H // 32, 2), max(W // 32, 2), generator=g)
    gt = F.interpolate(low, size=(H, W), mode='bicubic', align_corners=False).clamp(0, 1)
    lq = gt + torch.randn(B, 3, H, W, generator=g) * (sigma / 255.0)
    return {'lq': lq, 'gt': gt, 'ref': gt.clone()}


def randomize_gates(net, std=0.1, seed=0):
    """beta/gamma (NAFNet) and alpha (Restormer fusion blocks) are zero-initialised (blocks start as
    identities); give them N(0,std) so the benchmark exercises every kernel with non-trivial data."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k.endswith('beta') or k.endswith('gamma') or k.endswith('alpha'):
                p.copy_(torch.randn(p.shape, generator=g) * std)

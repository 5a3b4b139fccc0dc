"""tdr_crop_augment / DevicePairedAugmenter against the numpy restatement of the reference's host pipeline:
bit-exact (a gather, plus one fp32 multiply-add for the noise)."""
import random

import numpy as np
import pytest
import torch

from oracle import data_pipeline_oracle as DO

pytestmark = pytest.mark.gpu


def test_all_modes_and_offsets_bit_exact():
    from textualdegremoval_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    src = torch.rand(8, 3, 40, 52, generator=g)
    top = [0, 3, 8, 1, 0, 5, 7, 2]
    left = [0, 20, 7, 19, 1, 0, 11, 4]
    P = 32
    out = K.crop_augment(src.cuda(), torch.tensor(top, dtype=torch.int32).cuda(), torch.tensor(left, dtype=torch.int32).cuda(),
                         torch.arange(8, dtype=torch.int32).cuda(), P).cpu().numpy()
    for n in range(8):
        ref = DO.crop_augment(src[n].numpy(), top[n], left[n], P, n)
        assert np.array_equal(out[n], ref), n


def test_noise_synthesis_and_shared_parameters():
    from textualdegremoval_amd.data.device_pipeline import DevicePairedAugmenter
    rng = random.Random(7)
    aug = DevicePairedAugmenter({'gt_size': 64, 'geometric_augs': True, 'sigma_type': 'choice', 'sigma_range': [15, 25, 50]}, rng=rng)
    g = torch.Generator().manual_seed(1)
    gt = torch.rand(4, 3, 96, 80, generator=g).cuda()
    ref = torch.rand(4, 3, 96, 96, generator=g).cuda()
    gen = torch.Generator(device='cuda').manual_seed(5)
    out = aug(gt, ref=ref, generator=gen)
    p = aug.last
    # the same draws with the same seed, in the reference's order per sample
    rr = random.Random(7)
    for n in range(4):
        assert p['top'][n] == rr.randint(0, 96 - 64) and p['left'][n] == rr.randint(0, 80 - 64) and p['mode'][n] == rr.randint(0, 7)
        assert p['sigma'][n] == float(rr.choice([15, 25, 50]))
    noise = torch.randn(4, 3, 64, 64, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)).cpu().numpy()
    for n in range(4):
        clean = DO.crop_augment(gt[n].cpu().numpy(), p['top'][n], p['left'][n], 64, p['mode'][n])
        assert np.array_equal(out['gt'][n].cpu().numpy(), clean)
        lq = clean + noise[n] * np.float32(p['sigma'][n] / 255.0)
        assert np.abs(out['lq'][n].cpu().numpy() - lq).max() < 1e-6
        assert np.array_equal(out['ref'][n].cpu().numpy(), DO.crop_augment(ref[n].cpu().numpy(), 0, 0, 96, p['mode'][n]))
    res = (out['lq'] - out['gt']).cpu()
    for n in range(4):
        assert abs(res[n].std().item() - p['sigma'][n] / 255.0) < 0.05 * p['sigma'][n] / 255.0


def test_progressive_resize_hook_shapes():
    """the trainer's progressive-learning slicing (main_train_restoration_with_ref_input.py:240-270) feeds smaller patches and
    batches: the step re-captures its graphs per shape and keeps training."""
    from textualdegremoval_amd.models import create_model
    import bench
    from textualdegremoval_amd.utils.synthetic import synthetic_pair
    m = create_model(bench.make_opt(16, [1, 1, 1, 1], 128, False))
    data = {k: v.cuda() for k, v in synthetic_pair(4, 128, 128, seed=3).items()}
    losses = []
    it = 0
    for bs, size in ((4, 128), (4, 128), (4, 128), (2, 64), (2, 64), (2, 64), (4, 128)):
        it += 1
        x0 = y0 = (128 - size) // 2
        d = {'lq': data['lq'][:bs, :, x0:x0 + size, y0:y0 + size].contiguous(), 'gt': data['gt'][:bs, :, x0:x0 + size, y0:y0 + size].contiguous(),
             'ref': data['ref'][:bs, :, x0:x0 + size, y0:y0 + size].contiguous()}
        m.update_learning_rate(it, warmup_iter=-1)
        m.feed_train_data(d)
        m.optimize_parameters(it)
        losses.append(float(m.get_current_log()['l_pix']))
    assert all(np.isfinite(losses))

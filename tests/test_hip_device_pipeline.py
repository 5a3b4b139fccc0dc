"""tdr_crop_augment / DevicePairedAugmenter against vectors produced by the reference's own data/transforms.py
(tests/golden/transforms.npz, make_golden_transforms.py) and the pinned oracle: bit-exact (a gather, plus one fp32
multiply-add for the noise)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data_pipeline_oracle as DO

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transforms.npz'))


def _image(seed, h, w):
    return np.random.RandomState(seed).rand(h, w, 3).astype(np.float32)


@pytest.mark.parametrize('case', [tuple(int(v) for v in c) for c in G['cases']])
def test_augmenter_reproduces_the_reference_pipeline(case):
    """same python-`random` seed -> same draws, same pixels as paired_random_crop + random_augmentation of the reference"""
    from textualdegremoval_amd.data.device_pipeline import DevicePairedAugmenter
    seed, h, w, patch = case
    aug = DevicePairedAugmenter({'gt_size': patch, 'geometric_augs': True}, rng=random.Random(seed))
    chw = lambda a: torch.from_numpy(a.transpose(2, 0, 1).copy())[None].cuda()
    ref = torch.rand(1, 3, h + 9, w + 5).cuda()
    out = aug(chw(_image(seed, h, w)), lq=chw(_image(seed + 1000, h, w)), ref=ref)
    assert [aug.last['top'][0], aug.last['left'][0], aug.last['mode'][0]] == [int(v) for v in G[f'c{seed}_draws']]
    assert np.array_equal(out['gt'][0].cpu().numpy().transpose(1, 2, 0), G[f'c{seed}_gt'])
    assert np.array_equal(out['lq'][0].cpu().numpy().transpose(1, 2, 0), G[f'c{seed}_lq'])
    assert out['ref'] is ref                   # the WithRef datasets never crop or augment img_ref


def test_noise_vs_reference_statements():
    """lq = gt + randn * sigma / 255 against the dataset's own statements (restoration_dataset.py:465-476), mode 0, full image"""
    from textualdegremoval_amd import kernels as K
    img = torch.from_numpy(_image(30, 24, 20).transpose(2, 0, 1).copy())
    torch.manual_seed(78)
    randn = torch.randn(3, 24, 20)
    for tag in ('const', 'rand', 'choice'):
        sig = torch.tensor([float(G[f'noise_{tag}_sigma'])], dtype=torch.float32) / 255.0
        # the kernel crops squares: two overlapping 20x20 crops cover the 24x20 image
        for top in (0, 4):
            out = K.crop_augment(img[None].cuda(), torch.tensor([top], dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda(),
                                 torch.zeros(1, dtype=torch.int32).cuda(), 20, noise=randn[None, :, top:top + 20].contiguous().cuda(),
                                 sigma=sig.cuda())
            # (the kernel may contract the multiply-add into one FMA: half an ulp of 1.0 either way)
            assert np.abs(out[0].cpu().numpy() - G[f"noise_{tag}_out"][:, top:top + 20]).max() <= 1.3e-7


def test_small_images_are_reflect_padded_like_padding():
    from textualdegremoval_amd.data.device_pipeline import DevicePairedAugmenter
    aug = DevicePairedAugmenter({'gt_size': 32, 'geometric_augs': True}, rng=random.Random(3))
    gt = torch.rand(3, 3, 20, 27)
    out = aug(gt.cuda())
    for n in range(3):
        want = DO.crop_augment(gt[n].numpy(), aug.last['top'][n], aug.last['left'][n], 32, aug.last['mode'][n])
        assert np.array_equal(out['gt'][n].cpu().numpy(), want)



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
    assert out['ref'] is ref
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

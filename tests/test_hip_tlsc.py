"""Validation-path pieces of SURVEY 8 row f2 on the device against vectors produced by the REFERENCE classes on CPU
(tests/golden/make_golden_tlsc.py): TLSC `AvgPool2d` (models/archs/nafnet_local_arch.py:10-75), `NAFNetLocal`
(network_nafnet_guided_arch.py:756-768) and the float64 Y-channel SSIM `_ssim_cly` (metrics/psnr_ssim.py:184-222)."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tlsc.npz')


def test_oracle_ssim_cly_is_pinned_by_the_reference_formula():
    """CPU: the oracle's restatement against `_ssim_cly` executed from the reference file (float64: agreement to 1e-12)"""
    from oracle import metrics_oracle as MO
    g = np.load(GOLDEN)
    for i in range(int(g['ssim_n'])):
        assert abs(MO.ssim_cly(g[f'ssim{i}_a'], g[f'ssim{i}_b']) - float(g[f'ssim{i}_val'])) < 1e-12


@pytest.mark.gpu
def test_ssim_y64_kernel_vs_reference():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    g = np.load(GOLDEN)
    for i in range(int(g['ssim_n'])):
        got = K.ssim_y64(torch.from_numpy(g[f'ssim{i}_a']).cuda(), torch.from_numpy(g[f'ssim{i}_b']).cuda())
        # every field in double on both sides; the separable evaluation of the window differs from filter2D's dense one by double rounding
        assert abs(got - float(g[f'ssim{i}_val'])) < 1e-10, (i, got, float(g[f'ssim{i}_val']))


@pytest.mark.gpu
def test_calculate_ssim_y_channel_runs_in_double():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from oracle import metrics_oracle as MO
    from textualdegremoval_amd import metrics as M
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (48, 40, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.float64) + rng.normal(0, 6, a.shape), 0, 255).round().astype(np.uint8)
    got = M.calculate_ssim(a, b, crop_border=3, test_y_channel=True)
    assert abs(got - MO.calculate_ssim(a, b, 3, test_y_channel=True)) < 1e-9


@pytest.mark.gpu
def test_tlsc_avgpool_vs_reference_class():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd.models.archs.nafnet_local_arch import AvgPool2d
    g = np.load(GOLDEN)
    for i in range(int(g['pool_n'])):
        x = torch.from_numpy(g[f'pool{i}_x']).cuda()
        out = AvgPool2d(kernel_size=[int(v) for v in g[f'pool{i}_k']])(x)
        # the reference differences a float32 integral image (values up to k1*k2*|x|): 2e-5 covers ITS cancellation error
        assert out.shape == x.shape and (out.cpu() - torch.from_numpy(g[f'pool{i}_out'])).abs().max().item() < 2e-5, i
    # a kernel covering the whole map is the global mean
    x = torch.from_numpy(g['pool0_x']).cuda()
    out = AvgPool2d(kernel_size=[64, 64])(x)
    assert out.shape == (2, 3, 1, 1) and (out.cpu() - torch.from_numpy(g['pool0_x']).mean((2, 3), keepdim=True)).abs().max().item() < 1e-6
    # lazily fixed kernel: base_size / train_size semantics of :29-36
    p = AvgPool2d(base_size=(30, 42), train_size=(1, 3, 20, 28))
    p(x)
    assert list(p.kernel_size) == [20 * 30 // 20, 28 * 42 // 28]


@pytest.mark.gpu
@pytest.mark.parametrize('math', ['hx2', 'f32'])
def test_nafnet_local_vs_reference(math):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models.archs import define_network
    g = np.load(GOLDEN)
    prev = K.MATH
    K.set_math(math)
    try:
        net = define_network(dict(type='NAFNetLocal', img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1], dec_blk_nums=[1, 1],
                                  train_size=(1, 3, 32, 32)))
        assert [k for k, _ in net.named_parameters()] == [str(k) for k in g['nl_names']]
        net.load_state_dict({str(k): torch.from_numpy(g['nl_p_' + str(k)]) for k in g['nl_names']}, strict=True)
        net = net.cuda()
        # the kernels Local_Base.convert fixed in the reference, per U-Net level (modules() order: enc0, enc1, dec0 (level 1), dec1 (level 0), middle)
        ks = [tuple(int(v) for v in k) for k in g['nl_ksizes']]
        assert net.ksizes == [ks[0], ks[1], ks[4]] and ks[2] == ks[1] and ks[3] == ks[0]
        assert not net.training
        out = net(torch.from_numpy(g['nl_x']).cuda())
        assert not out.requires_grad
        assert (out.cpu() - torch.from_numpy(g['nl_out'])).abs().max().item() < 1e-4
    finally:
        K.set_math(prev)
    with pytest.raises(NotImplementedError):
        define_network(dict(type='NAFNetLocal', img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1], dec_blk_nums=[1], fast_imp=True))

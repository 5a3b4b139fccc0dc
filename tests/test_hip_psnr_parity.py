"""PSNR parity (north_star: within 1e-3 dB of the reference CPU path): the validation path of the step API
(`validation` -> nonpad_test -> tensor2img -> calculate_psnr) on the HIP network against the same metric computed
from the reference's own output (golden fixtures produced by running the reference) and from the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as NO
from oracle import restormer_ref_oracle as RO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _opt(net, metrics=True):
    return {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 1, 'dist': False, 'is_train': True, 'network_g': net, 'path': {},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                  'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'scale': 1,
        'val': {'metrics': {'psnr': {'type': 'calculate_psnr', 'crop_border': 0, 'test_y_channel': False}}},
    }


CASES = [
    ('nafnet', 'net_w8_256_b2', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]), 1234),
    ('restormer', 'restormer_d8_128_biasfree_b2', dict(LayerNorm_type='BiasFree', num_blocks=[1, 2, 1, 1]), 4321),
]


@pytest.mark.parametrize('arch,name,kw,seed0', CASES)
def test_validation_psnr_matches_reference_output(arch, name, kw, seed0):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd.metrics import calculate_psnr, tensor2img
    from textualdegremoval_amd.models import create_model
    g = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    seed = int(g['seed'])
    if arch == 'nafnet':
        cfg = NO.default_cfg(**kw)
        P = NO.synth_params(cfg, seed=seed)
        net = dict(type='NAFNetRefFusion', width=cfg['width'], nf=cfg['nf'], enc_blk_nums=cfg['enc_blk_nums'],
                   dec_blk_nums=cfg['dec_blk_nums'], middle_blk_num=cfg['middle_blk_num'], ext_n_blocks=cfg['ext_n_blocks'],
                   reffusion_n_blocks=cfg['reffusion_n_blocks'])
    else:
        cfg = RO.default_cfg(**kw)
        P = RO.synth_params(cfg, seed=seed)
        net = dict(type='RestormerRefFusion', **cfg)
    model = create_model(_opt(net))
    model.net_g.load_state_dict(P, strict=True)
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=seed0 + seed)
    ref_out = torch.from_numpy(g['out'])                 # the reference network's output on the same inputs
    for b in range(lq.shape[0]):                         # the reference validates image by image (batch 1 loaders)
        data = [{'lq': lq[b:b + 1], 'gt': gt[b:b + 1], 'ref': ref[b:b + 1]}]
        psnr_hip = model.validation(data, 0, None, save_img=False, rgb2bgr=True, use_image=True)
        want = calculate_psnr(tensor2img(ref_out[b:b + 1]), tensor2img(gt[b:b + 1]))
        assert abs(psnr_hip - want) < 1e-3, (b, psnr_hip, want)
        # float-domain PSNR (use_image=False), where nothing is hidden by the uint8 rounding
        pf_hip = calculate_psnr(model.output.clamp(0, 1).cpu(), gt[b:b + 1])
        pf_ref = calculate_psnr(ref_out[b:b + 1].clamp(0, 1), gt[b:b + 1])
        assert abs(pf_hip - pf_ref) < 1e-3, (b, pf_hip, pf_ref)


def test_validation_with_ssim_metric_through_the_step_api():
    """`val.metrics` with both calculate_psnr and calculate_ssim (basicsr-style option block): the SSIM of the HIP network's
    uint8 output against the oracle's SSIM of the reference network's output on the same inputs."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from oracle import metrics_oracle as MO
    from textualdegremoval_amd.metrics import tensor2img
    from textualdegremoval_amd.models import create_model
    g = np.load(os.path.join(GOLDEN, 'net_w8_256_b2.npz'), allow_pickle=False)
    seed = int(g['seed'])
    cfg = NO.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    net = dict(type='NAFNetRefFusion', width=cfg['width'], nf=cfg['nf'], enc_blk_nums=cfg['enc_blk_nums'],
               dec_blk_nums=cfg['dec_blk_nums'], middle_blk_num=cfg['middle_blk_num'], ext_n_blocks=cfg['ext_n_blocks'],
               reffusion_n_blocks=cfg['reffusion_n_blocks'])
    opt = _opt(net)
    opt['val']['metrics']['ssim'] = {'type': 'calculate_ssim', 'crop_border': 0, 'test_y_channel': False}
    model = create_model(opt)
    model.net_g.load_state_dict(NO.synth_params(cfg, seed=seed), strict=True)
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=1234 + seed)
    ref_out = torch.from_numpy(g['out'])
    last = model.validation([{'lq': lq[:1], 'gt': gt[:1], 'ref': ref[:1]}], 0, None, save_img=False, rgb2bgr=True, use_image=True)
    want = MO.calculate_ssim(tensor2img(ref_out[:1]), tensor2img(gt[:1]), 0)
    assert abs(model.metric_results['ssim'] - want) < 2e-4, (model.metric_results, want)
    assert last == model.metric_results['ssim'] and model.metric_results['psnr'] > 0

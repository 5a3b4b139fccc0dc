"""GPU parity of the train/step API: 3 iterations of RefGuidedImageCleanModel.optimize_parameters
against the trajectory recorded from the reference's own step API on CPU
(tests/golden/trajectory.npz): losses, LRs, parameter checksums, final output."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def make_opt():
    return {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 1, 'dist': False, 'is_train': True,
        'network_g': dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=[1, 1, 1, 1], dec_blk_nums=[1, 1, 1, 1],
                          middle_blk_num=1, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]),
        'path': {},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                  'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'val': {}, 'scale': 1,
    }


def test_three_step_trajectory_vs_reference_step_api():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd.models import create_model
    g = np.load(os.path.join(GOLDEN, 'trajectory.npz'))
    model = create_model(make_opt())
    cfg = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    model.net_g.load_state_dict(O.synth_params(cfg, seed=3), strict=True)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1234 + 3)
    for it in range(1, 4):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref})
        model.optimize_parameters(it)
        loss = model.get_current_log()['l_pix']
        lrs = model.get_current_learning_rate()
        assert abs(loss - g['losses'][it - 1]) < 3e-6, (it, loss, g['losses'][it - 1])
        assert np.allclose(lrs, g['lrs'][it - 1], rtol=0, atol=1e-12)
        assert torch.equal(model.ref_in, model.ref)
    sd = model.net_g.state_dict()
    psum = np.array([sd[k].double().sum().item() for k in sd])
    assert np.allclose(psum, g['psum'], rtol=0, atol=1e-3)
    assert (model.output.cpu() - torch.from_numpy(g['final_out'])).abs().max().item() < 1e-4


def test_checkpoint_round_trip(tmp_path):
    """save()/resume_training() keep the reference's file layout: {'params': sd} and .state with AdamW keys."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd.models import create_model
    opt = make_opt()
    opt['path'] = {'models': str(tmp_path), 'training_states': str(tmp_path)}
    model = create_model(opt)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=5)
    model.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref})
    model.optimize_parameters(1)
    model.save(0, 1)
    ck = torch.load(os.path.join(str(tmp_path), 'net_g_1.pth'))
    assert list(ck['params'].keys()) == list(model.net_g.state_dict().keys())
    st = torch.load(os.path.join(str(tmp_path), '1.state'))
    s0 = st['optimizers'][0]['state'][0]
    assert set(s0.keys()) >= {'step', 'exp_avg', 'exp_avg_sq'} and len(st['optimizers'][0]['param_groups']) == 2
    model2 = create_model(make_opt() | {'path': {'pretrain_network_g': os.path.join(str(tmp_path), 'net_g_1.pth'),
                                                 'strict_load_g': True}})
    model2.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref})
    model2.optimize_parameters(1)          # materialise optimiser state, then overwrite it
    model2.resume_training(st)
    model.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref}); model.optimize_parameters(2)
    model2.net_g.load_state_dict(ck['params']); model2.optimize_parameters(2)
    a = model.get_current_log()['l_pix']; b = model2.get_current_log()['l_pix']
    assert abs(a - b) < 1e-7


def test_step_with_dino_window_match(tmp_path):
    """ref larger than lq: the frozen ViT matcher (path.pretrain_dino) picks the window; ref_in must equal the oracle's."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import torch.nn.functional as F
    from oracle import dino_oracle as D
    from textualdegremoval_amd.models import create_model
    sd = D.synth_vit_params(192, 1, 12, seed=21)                      # 12 heads like ViT-B (head dim 16), one block
    ck = os.path.join(str(tmp_path), 'dino.pth')
    torch.save(sd, ck)
    opt = make_opt()
    opt['path'] = {'pretrain_dino': ck}
    model = create_model(opt)
    g = torch.Generator().manual_seed(5)
    clean = F.interpolate(torch.rand(1, 3, 16, 16, generator=g), size=(256, 256), mode='bicubic').clamp(0, 1)
    gt = clean[:, :, 64:192, 32:160].contiguous()                      # window (row 2, col 1) of the stride-32 grid
    lq = gt + torch.randn(1, 3, 128, 128, generator=g) * (15 / 255)
    o_ref_in, o_idx, o_corr = D.match_reference_window(sd, lq, clean, heads=12)
    for it in (1, 2, 3, 4):                                            # eager, eager, capture, replay
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data({'lq': lq, 'gt': gt, 'ref': clean})
        model.optimize_parameters(it)
        assert int(model.match_index[0]) == int(o_idx[0]) == 2 * 5 + 1
        assert torch.equal(model.ref_in.cpu(), o_ref_in)
        # similarities: the matcher's Linears run on ONE fp16 product by default (dino.py; only the index leaves the sub-graph);
        # tests/test_hip_dino.py pins 1e-5 for the split / exact arithmetic
        assert (model.match_corr.cpu() - o_corr).abs().max().item() < 5e-3
    assert np.isfinite(model.get_current_log()['l_pix'])

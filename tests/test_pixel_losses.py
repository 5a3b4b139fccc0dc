"""Pixel criteria (losses/losses.py:26-122 of the reference).  CPU: oracle/losses_oracle.py against the golden values and
gradients of the reference itself (tests/golden/losses.npz, made by make_golden_losses.py).  GPU: tdr_pixel_loss through
textualdegremoval_amd.losses and through the train step with each `pixel_opt.type`."""
import os

import numpy as np
import pytest
import torch

from oracle import losses_oracle as LO
from oracle import nafnet_ref_oracle as NO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = [('l1', 'L1Loss', dict(loss_weight=0.7)), ('mse', 'MSELoss', dict(loss_weight=1.3)),
         ('charbonnier', 'CharbonnierLoss', dict(loss_weight=5.0, eps=1e-3)), ('charbonnier_eps2', 'CharbonnierLoss', dict(eps=0.05)),
         ('psnr', 'PSNRLoss', dict(loss_weight=0.5)), ('psnr_y', 'PSNRLoss', dict(loss_weight=1.0, toY=True))]


def _oracle(name, kw, p, t):
    if name.startswith('charbonnier'):
        return LO.charbonnier(p, t, kw.get('eps', 1e-3))
    if name.startswith('psnr'):
        return LO.psnr(p, t, kw.get('loss_weight', 1.0), toY=kw.get('toY', False))
    return LO.KINDS[name](p, t, kw.get('loss_weight', 1.0))


@pytest.mark.parametrize('name,cls,kw', CASES)
def test_oracle_matches_reference_golden(name, cls, kw):
    g = np.load(os.path.join(GOLDEN, 'losses.npz'))
    loss, grad = _oracle(name, kw, g['pred'], g['target'])
    assert abs(loss - float(g[name + '_loss'])) <= 2e-6 * max(1.0, abs(loss))
    ref = g[name + '_grad'].astype(np.float64)
    if name == 'l1':            # sign(): compare where the residual is not a float32 tie
        assert np.array_equal(np.sign(grad), np.sign(ref))
    assert np.abs(grad - ref).max() <= 2e-5 * np.abs(ref).max()


def test_host_tensors_keep_the_torch_expressions():
    from textualdegremoval_amd import losses as L
    g = np.load(os.path.join(GOLDEN, 'losses.npz'))
    p, t = torch.tensor(g['pred']), torch.tensor(g['target'])
    for name, cls, kw in CASES:
        v = getattr(L, cls)(**kw)(p, t)
        assert abs(float(v) - float(g[name + '_loss'])) <= 2e-6 * max(1.0, abs(float(v)))
    with pytest.raises(ValueError):
        L.MSELoss(reduction='avg')
    assert L.L1Loss(reduction='sum').step_kind() is None and L.MSELoss().step_kind() is not None


WEIGHTED = [('l1_w1_mean', 'L1Loss', dict(loss_weight=0.7), 'w1'), ('l1_w3_mean', 'L1Loss', {}, 'w3'),
            ('mse_w1_mean', 'MSELoss', dict(loss_weight=2.0), 'w1'), ('mse_w3_sum', 'MSELoss', dict(reduction='sum'), 'w3'),
            ('l1_sum', 'L1Loss', dict(reduction='sum'), None), ('l1_w1_none', 'L1Loss', dict(reduction='none'), 'w1')]


@pytest.mark.parametrize('name,cls,kw,wk', WEIGHTED)
def test_weighted_and_non_mean_reductions_follow_weight_reduce_loss(name, cls, kw, wk):
    """losses/loss_util.py:25-54 through the reference's own classes: a one-channel weight divides by weight.sum() * C"""
    from textualdegremoval_amd import losses as L
    g = np.load(os.path.join(GOLDEN, 'losses.npz'))
    p = torch.tensor(g['pred'], requires_grad=True)
    v = getattr(L, cls)(**kw)(p, torch.tensor(g['target']), weight=None if wk is None else torch.tensor(g[wk]))
    assert np.allclose(v.detach().numpy().astype(np.float64), g[name + '_loss'], rtol=3e-6, atol=1e-9)
    v.sum().backward()
    assert np.allclose(p.grad.numpy(), g[name + '_grad'], rtol=1e-5, atol=1e-9)
    if wk == 'w1':
        with pytest.raises(AssertionError):
            getattr(L, cls)(**kw)(p, torch.tensor(g['target']), weight=torch.tensor(g['w3'][:, :2]))


@pytest.mark.gpu
@pytest.mark.parametrize('name,cls,kw', CASES)
def test_hip_pixel_loss_matches_golden_and_oracle(name, cls, kw):
    from textualdegremoval_amd import losses as L
    g = np.load(os.path.join(GOLDEN, 'losses.npz'))
    p = torch.tensor(g['pred'], device='cuda', requires_grad=True)
    t = torch.tensor(g['target'], device='cuda')
    loss = getattr(L, cls)(**kw)(p, t)
    loss.backward()
    loss = loss.detach()
    want, ograd = _oracle(name, kw, g['pred'], g['target'])
    assert abs(float(loss) - float(g[name + '_loss'])) <= 2e-6 * max(1.0, abs(want))
    assert abs(float(loss) - want) <= 2e-6 * max(1.0, abs(want))
    got = p.grad.cpu().numpy().astype(np.float64)
    ref = g[name + '_grad'].astype(np.float64)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    assert np.abs(got - ograd).max() <= 2e-5 * np.abs(ograd).max()


@pytest.mark.gpu
def test_hip_pixel_loss_larger_batch_and_scale():
    from textualdegremoval_amd import kernels as K
    rng = np.random.default_rng(5)
    t = rng.random((70, 3, 33, 17)).astype(np.float32)         # > 64 images: the strided finish
    p = (t + rng.normal(0, 0.05, t.shape)).astype(np.float32)
    tp, tt = torch.tensor(p, device='cuda'), torch.tensor(t, device='cuda')
    for kind, name, kw in [(K.LOSS_PSNR, 'psnr', {}), (K.LOSS_PSNR_Y, 'psnr', {'toY': True}), (K.LOSS_MSE, 'mse', {})]:
        loss, d = K.pixel_loss(kind, tp, tt, 1.0, 0.0, grad_scale=1024.0)
        want, og = (LO.psnr(p, t, 1.0, **kw) if name == 'psnr' else LO.mse(p, t, 1.0))
        assert abs(float(loss) - want) <= 2e-6 * max(1.0, abs(want))
        assert np.abs(d.cpu().numpy() / 1024.0 - og).max() <= 2e-5 * np.abs(og).max()


def _opt(net, pixel_opt):
    return {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 1, 'dist': False, 'is_train': True, 'network_g': net, 'path': {},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': pixel_opt, 'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'scale': 1, 'val': {},
    }


@pytest.mark.gpu
@pytest.mark.parametrize('name,cls,kw', [c for c in CASES if c[0] != 'charbonnier_eps2'])
def test_train_step_with_each_criterion(name, cls, kw):
    """the fused step with `pixel_opt.type` = each criterion: loss value and parameter gradients against the oracle network
    differentiated through the oracle criterion (same forward, cotangent = the criterion's gradient)"""
    from textualdegremoval_amd.models import create_model
    cfg = NO.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    net = dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=cfg['enc_blk_nums'], dec_blk_nums=cfg['dec_blk_nums'],
               middle_blk_num=cfg['middle_blk_num'], ext_n_blocks=cfg['ext_n_blocks'], reffusion_n_blocks=cfg['reffusion_n_blocks'])
    P = NO.synth_params(cfg, seed=3)
    lq, gt, ref = NO.synth_pair(2, 128, 128, seed=11)
    os.environ['TDR_GRAPH'] = '0'
    try:
        model = create_model(_opt(net, dict(type=cls, **kw)))
        model.net_g.load_state_dict(P, strict=True)
        model.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref})
        model.ref_in = model._match_reference_window()
        model.optimize_parameters(1)
        l_pix = float(model.log_dict['l_pix']) if hasattr(model, 'log_dict') else None
    finally:
        os.environ.pop('TDR_GRAPH', None)
    # oracle: forward with torch autograd on the oracle network, criterion gradient from the numpy oracle as the cotangent
    Pt = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = NO.nafnet_ref_forward(Pt, cfg, lq, ref)
    want, og = _oracle(name, kw, out.detach().numpy(), gt.numpy())
    assert l_pix is not None and abs(l_pix - want) <= 1e-4 * max(1.0, abs(want)), (l_pix, want)
    out.backward(torch.tensor(og, dtype=torch.float32))
    bad = []
    for k, p in model.net_g.named_parameters():
        if Pt[k].grad is None or p.grad is None:
            continue
        r = Pt[k].grad.float()
        e = (p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-12)
        if e > 5e-3:
            bad.append((k, float(e)))
    assert not bad, bad[:5]

"""Pins oracle/restormer_ref_oracle.py against golden vectors produced by running the reference's own
Restormer-ref classes (tests/golden/make_golden_restormer.py).  Tolerances: 2e-5 max-abs on O(1) fp32
activations (reference target is 1e-4), exact equality for integer indices."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as NO
from oracle import restormer_ref_oracle as RO



def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def _params(g, tag, pre=''):
    return {pre + str(k): T(g[f'{tag}_p_{k}']).requires_grad_(True) for k in g[tag + '_names']}


def _check(g, tag, x, y, P, pre='', tol=2e-5):
    y.backward(T(g[tag + '_go']))
    assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < tol
    assert np.abs(x.grad.numpy() - g[tag + '_gx']).max() < tol
    for k, p in P.items():
        ref = g[f'{tag}_g_{k[len(pre):]}']
        assert np.abs(p.grad.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize('kind', ['BiasFree', 'WithBias'])
def test_layernorm_variants(golden_dir, kind):
    g = load(golden_dir, 'restormer_per_op')
    tag = 'ln_' + kind
    P = _params(g, tag, 'n.')
    x = T(g[tag + '_x']).requires_grad_(True)
    _check(g, tag, x, RO.layernorm(x, P, 'n.', kind), P, 'n.')


@pytest.mark.parametrize('b', [0, 1])
def test_gdfn(golden_dir, b):
    g = load(golden_dir, 'restormer_per_op')
    tag = f'gdfn_b{b}'
    P = _params(g, tag, 'f.')
    x = T(g[tag + '_x']).requires_grad_(True)
    _check(g, tag, x, RO.gdfn(x, P, 'f.'), P, 'f.')


@pytest.mark.parametrize('b', [0, 1])
def test_mdta(golden_dir, b):
    g = load(golden_dir, 'restormer_per_op')
    tag = f'mdta_b{b}'
    P = _params(g, tag, 'a.')
    x = T(g[tag + '_x']).requires_grad_(True)
    _check(g, tag, x, RO.mdta(x, P, 'a.', 2), P, 'a.')


@pytest.mark.parametrize('kind', ['BiasFree', 'WithBias'])
def test_transformer_block(golden_dir, kind):
    g = load(golden_dir, 'restormer_per_op')
    tag = 'tblock_' + kind
    P = _params(g, tag, 'b.')
    x = T(g[tag + '_x']).requires_grad_(True)
    _check(g, tag, x, RO.transformer_block(x, P, 'b.', 4, kind), P, 'b.', tol=5e-5)


def test_fusion_block(golden_dir):
    g = load(golden_dir, 'restormer_per_op')
    P = _params(g, 'fblock', 'b.')
    x = T(g['fblock_x']).requires_grad_(True)
    _check(g, 'fblock', x, RO.fusion_block(x, P, 'b.', 2, 'WithBias'), P, 'b.', tol=5e-5)


def test_down_up_sample(golden_dir):
    g = load(golden_dir, 'restormer_per_op')
    for tag, fn in (('down', RO.downsample), ('up', RO.upsample)):
        P = _params(g, tag, 'r.')
        x = T(g[tag + '_x']).requires_grad_(True)
        _check(g, tag, x, fn(x, P, 'r.'), P, 'r.')


CASES = [('restormer_d8_128', dict()),
         ('restormer_d8_128_biasfree_b2', dict(LayerNorm_type='BiasFree', num_blocks=[1, 2, 1, 1])),
         ('restormer_d8_64_wrap_bias', dict(bias=True)),
         ('restormer_d16_120x100_pad', dict(dim=16, nf=16))]


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_forward_backward(golden_dir, name, kw):
    g = load(golden_dir, name)
    cfg = RO.default_cfg(**kw)
    seed = int(g['seed'])
    P = {k: v.requires_grad_(True) for k, v in RO.synth_params(cfg, seed=seed).items()}
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=4321 + seed)
    out, aux = RO.restormer_ref_forward(P, cfg, lq, ref, return_aux=True)
    assert np.array_equal(aux['index'].numpy(), g['index'][..., 0] if g['index'].ndim == 3 else g['index'])
    assert np.array_equal(aux['index_all'].numpy(), g['index_all'])
    assert np.abs(aux['soft_att'].detach().numpy()[:, 0] - g['soft_att']).max() < 1e-5
    assert np.abs(out.detach().numpy() - g['out']).max() < 2e-5
    loss = NO.l1_loss(out, gt)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    loss.backward()
    for i in range(4):          # reference calls transfer coarse->fine (x1..x8); oracle list is fine->coarse
        w = aux['warp'][3 - i].detach().double()
        st = np.array([w.sum().item(), w.abs().sum().item(), (w * w).sum().item()])
        assert np.allclose(st, g[f'warp{i}_stats'], rtol=1e-5, atol=1e-4), i
    assert g['has_grad'].all()
    gn = np.array([(p.grad.double().norm().item() if p.grad is not None else 0.0) for p in P.values()])
    assert np.allclose(gn, g['grad_norm'], rtol=2e-3, atol=2e-6)
    assert abs(np.sqrt((gn ** 2).sum()) - float(g['total_grad_norm'])) < 1e-4 * float(g['total_grad_norm'])

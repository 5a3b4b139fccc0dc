"""Pins oracle/drsformer_ref_oracle.py against golden vectors produced by running the reference's own DRSformer-ref classes
(tests/golden/make_golden_drsformer.py).  Tolerances: 2e-5 max-abs on O(1) fp32 activations, exact integer indices."""
import os

import numpy as np
import pytest
import torch

from oracle import drsformer_ref_oracle as DO
from oracle import nafnet_ref_oracle as NO


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_reference_defects_are_recorded(golden_dir):
    g = load(golden_dir, 'drsformer_defects')
    assert 'functools' in str(g['R5_no_functools']) and 'index out of range' in str(g['R1_pyramid_index'])


def _run(g, tag, fn, pre, **kw):
    P = {pre + str(k): T(g[f'{tag}_p_{k}']).requires_grad_(True) for k in g[tag + '_names']}
    x = T(g[tag + '_x']).requires_grad_(True)
    y = fn(x, P, pre, **kw)
    y.backward(T(g[tag + '_go']))
    assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 2e-5
    assert np.abs(x.grad.numpy() - g[tag + '_gx']).max() < 2e-5
    for k, p in P.items():
        ref = g[f'{tag}_g_{k[len(pre):]}']
        assert np.abs(p.grad.numpy().reshape(ref.shape) - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize('tag,heads', [('tksa_a', 2), ('tksa_b', 1), ('tksa_c', 4)])
def test_top_k_sparse_attention(golden_dir, tag, heads):
    _run(load(golden_dir, 'drsformer_per_op'), tag, DO.tksa, 'a.', heads=heads)


@pytest.mark.parametrize('tag', ['msfn_a', 'msfn_b'])
def test_mixed_scale_feed_forward(golden_dir, tag):
    _run(load(golden_dir, 'drsformer_per_op'), tag, DO.msfn, 'f.')


CASES = [('drsformer_d8_64', dict()),
         ('drsformer_d8_128_b2_biasfree', dict(LayerNorm_type='BiasFree', num_blocks=[1, 1, 2, 1])),
         ('drsformer_d16_100x72_pad', dict(dim=16, nf=16, bias=True))]


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_forward_backward(golden_dir, name, kw):
    g = load(golden_dir, name)
    cfg = DO.default_cfg(**kw)
    seed = int(g['seed'])
    P0 = DO.synth_params(cfg, seed=seed)
    assert [str(n) for n in g['names']] == list(P0.keys())
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=8765 + seed)
    with torch.no_grad():
        out, aux = DO.drsformer_ref_forward(P0, cfg, lq, ref, return_aux=True)
    assert np.array_equal(aux['index_all'].numpy(), g['index_all'])
    assert np.abs(out.numpy() - g['out']).max() < 2e-5
    _, loss, grads = DO.loss_and_grads(P0, cfg, lq, ref, gt)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    names = list(P0.keys())
    unused = [n for n, h in zip(names, g['has_grad']) if not h]
    assert unused and all(n.startswith(DO.UNUSED) for n in unused)                      # R6
    assert set(names) - set(unused) == set(grads.keys())
    gn = np.array([grads[n].double().norm().item() if n in grads else 0.0 for n in names])
    assert np.allclose(gn, g['grad_norm'], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize('tag', ['mefc_a', 'mefc_b'])
def test_mefc_subnet(golden_dir, tag):
    """`subnet` (:522-548): gating MLP + softmax over the 8 candidate operations, 4 weighted steps."""
    _run(load(golden_dir, 'drsformer_mefc'), tag, DO.mefc_subnet, 's.')


@pytest.mark.parametrize('name,kw', [('drsformer_full_d8_64', dict()), ('drsformer_full_d8_128_b2', dict(LayerNorm_type='BiasFree'))])
def test_full_class_with_mefc(golden_dir, name, kw):
    g = load(golden_dir, name)
    cfg = DO.default_cfg(**kw)
    seed = int(g['seed'])
    P0 = DO.full_synth_params(cfg, seed=seed)
    assert [str(n) for n in g['names']] == list(P0.keys())
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=8765 + seed)
    P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    out = DO.drsformer_full_forward(P, cfg, lq, ref)
    assert np.abs(out.detach().numpy() - g['out']).max() < 2e-5
    loss = (out - gt).abs().mean()
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    loss.backward()
    assert g['has_grad'].all()
    gn = np.array([p.grad.double().norm().item() for p in P.values()])
    assert np.allclose(gn, g['grad_norm'], rtol=2e-3, atol=2e-6)

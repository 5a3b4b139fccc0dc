"""Pins oracle/promptir_ref_oracle.py against golden vectors produced by running the reference's own PromptIR-ref classes
(tests/golden/make_golden_promptir.py).  Tolerances: 2e-5 max-abs on O(1) fp32 activations (reference target 1e-4), exact
equality for integer indices."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as NO
from oracle import promptir_ref_oracle as PO


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_shipped_configuration_raises_in_the_reference(golden_dir):
    """R4: decoder=False (the reference YAML's value) does not run -- recorded from the reference itself."""
    msg = str(load(golden_dir, 'promptir_decoder_false')['decoder_false_raises'])
    assert 'expected input' in msg and '192 channels' in msg
    with pytest.raises(ValueError):
        PO.promptir_ref_forward({}, PO.default_cfg(decoder=False), torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))


@pytest.mark.parametrize('tag', ['same', 'down', 'up'])
def test_prompt_gen_block(golden_dir, tag):
    g = load(golden_dir, 'promptir_prompt_block')
    P = {'p.' + k: T(g[f'{tag}_p_{k}']).requires_grad_(True)
         for k in ('prompt_param', 'linear_layer.weight', 'linear_layer.bias', 'conv3x3.weight')}
    x = T(g[tag + '_x']).requires_grad_(True)
    y = PO.prompt_gen(x, P, 'p.')
    y.backward(T(g[tag + '_go']))
    assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 2e-5
    assert np.abs(x.grad.numpy() - g[tag + '_gx']).max() < 2e-5
    for k, p in P.items():
        ref = g[f'{tag}_g_{k[2:]}']
        assert np.abs(p.grad.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k


CASES = [('promptir_d48_64', dict()),
         ('promptir_d48_128_b2_biasfree', dict(LayerNorm_type='BiasFree', num_blocks=[1, 1, 2, 1])),
         ('promptir_d48_100x72_pad', dict(bias=True))]


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_forward_backward(golden_dir, name, kw):
    g = load(golden_dir, name)
    cfg = PO.default_cfg(**kw)
    seed = int(g['seed'])
    P0 = PO.synth_params(cfg, seed=seed)
    assert [str(n) for n in g['names']] == list(P0.keys())
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=8765 + seed)
    with torch.no_grad():
        out, aux = PO.promptir_ref_forward(P0, cfg, lq, ref, return_aux=True)
    assert np.array_equal(aux['index'].numpy(), g['index'][..., 0] if g['index'].ndim == 3 else g['index'])
    assert np.array_equal(aux['index_all'].numpy(), g['index_all'])
    assert np.abs(out.numpy() - g['out']).max() < 2e-5
    out2, loss, grads = PO.loss_and_grads(P0, cfg, lq, ref, gt)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    names = list(P0.keys())
    unused = [n for n, h in zip(names, g['has_grad']) if not h]
    assert unused and all(n.startswith(PO.UNUSED) for n in unused)
    assert set(names) - set(unused) == set(grads.keys())
    gn = np.array([grads[n].double().norm().item() if n in grads else 0.0 for n in names])
    assert np.allclose(gn, g['grad_norm'], rtol=2e-3, atol=2e-6)
    assert abs(np.sqrt((gn ** 2).sum()) - float(g['total_grad_norm'])) < 1e-4 * float(g['total_grad_norm'])

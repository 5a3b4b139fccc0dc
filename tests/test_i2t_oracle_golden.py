"""Pins oracle/i2t_oracle.py against tests/golden/i2t_*.npz: the CLIP ViT tokens against
`transformers.CLIPVisionModel` (third-party; version recorded in the fixture), Mapper and the injected
cross-attention against the reference's own definitions (tests/golden/make_golden_i2t.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import i2t_oracle as IO


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_clip_vision_tokens(golden_dir, tag):
    g = load(golden_dir, 'i2t_clip')
    hidden, inter, layers, heads, image = [int(v) for v in g[tag + '_cfg']]
    sd = IO.synth_clip_params(hidden, inter, layers, 14, image, seed=ord(tag))
    out = IO.clip_vision_tokens(sd, T(g[tag + '_x']), heads, act=str(g[tag + '_act']))
    assert np.abs(out.numpy() - g[tag + '_out']).max() < 2e-5


def test_mapper_forward_backward(golden_dir):
    g = load(golden_dir, 'i2t_mapper')
    din, dout, words, B, Tn = [int(v) for v in g['cfg']]
    P = {k: v.requires_grad_(True) for k, v in IO.synth_mapper_params(din, 1280, dout, words, seed=5).items()}
    out = IO.mapper_forward(P, T(g['emb']), words)
    assert np.abs(out.detach().numpy() - g['out']).max() < 2e-5
    (out * T(g['go'])).sum().backward()
    names = [str(k) for k in g['names']]
    gn = np.array([P[k].grad.double().norm().item() for k in names])
    assert np.allclose(gn, g['grad_norm'], rtol=1e-3, atol=1e-6)


def test_clean_mapper_forward_backward(golden_dir):
    """oracle CleanMapper against the reference class (scripts/train/main_train_tr_mapping.py:84-120, executed from the reference
    file by make_golden_i2t.py::clean_mapper_case): output, gradient w.r.t. the input words, every parameter-gradient norm"""
    g = load(golden_dir, 'i2t_clean_mapper')
    din, dout, words, B = [int(v) for v in g['cfg']]
    P = {k: v.requires_grad_(True) for k, v in IO.synth_clean_mapper_params(din, 1280, dout, words, seed=6).items()}
    inj = T(g['inj']).requires_grad_(True)
    out = IO.clean_mapper_forward(P, inj, words)
    assert np.abs(out.detach().numpy() - g['out']).max() < 2e-5
    (out * T(g['go'])).sum().backward()
    assert np.abs(inj.grad.numpy() - g['dinj']).max() < 2e-5 * max(1.0, np.abs(g['dinj']).max())
    names = [str(k) for k in g['names']]
    gn = np.array([P[k].grad.double().norm().item() for k in names])
    assert np.allclose(gn, g['grad_norm'], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize('tag', ['x', 's'])
def test_injected_cross_attention(golden_dir, tag):
    g = load(golden_dir, 'i2t_xattn')
    dq, dc, inner, heads, B, Tq, Tk = [int(v) for v in g[tag + '_cfg']]
    P = {k[len(tag) + 3:]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith(tag + '_p_')}
    hid = T(g[tag + '_hid']).requires_grad_(True)
    ctx = T(g[tag + '_ctx']).requires_grad_(True) if Tk else None
    out = IO.cross_attention(P, hid, ctx, heads, (inner // heads) ** -0.5)
    assert np.abs(out.detach().numpy() - g[tag + '_out']).max() < 1e-5
    (out * T(g[tag + '_go'])).sum().backward()
    assert np.abs(hid.grad.numpy() - g[tag + '_ghid']).max() < 1e-5
    if Tk:
        assert np.abs(ctx.grad.numpy() - g[tag + '_gctx']).max() < 1e-5
    for k, p in P.items():
        key = f'{tag}_g_{k}'
        if key in g.files:
            assert np.abs(p.grad.numpy() - g[key]).max() < 1e-4 * max(1.0, np.abs(g[key]).max()), k

"""oracle/sfnet_oracle.py against the reference-generated goldens (tests/golden/sfnet.npz, made by tests/golden/make_golden_sfnet.py from
the reference's un-guided SFNet, dynamic_filter and the layers under them): outputs at the three scales, every parameter gradient's norm
and maximum, a few whole gradients, the BatchNorm buffers after a training-mode forward pass; dynamic_filter alone with the gradient
w.r.t. its input.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import sfnet_oracle as SO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sfnet.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLDEN, allow_pickle=False)


@pytest.mark.parametrize('tag', ['net_r2', 'net_r1_rect'])
def test_whole_network_forward_backward_and_buffers(g, tag):
    num_res, seed, n, h, w = (int(v) for v in g[tag + '_cfg'])
    sd = SO.synth_state(num_res, seed)
    P = {k: (v.clone().requires_grad_(True) if not SO.is_buffer(k) else v.clone()) for k, v in sd.items()}
    x = torch.from_numpy(g[tag + '_x'])
    bufs = {}
    outs = SO.sfnet_forward(P, x, num_res, bufs)
    for i, o in enumerate(outs):
        want = torch.from_numpy(g[f'{tag}_out{i}'])
        assert o.shape == want.shape and (o.detach() - want).abs().max().item() < 2e-5, (i, (o.detach() - want).abs().max().item())
    sum((o * torch.from_numpy(g[f'{tag}_go{i}'])).sum() for i, o in enumerate(outs)).backward()
    names = [str(s) for s in g[tag + '_names']]
    assert names == [k for k in sd if not SO.is_buffer(k)]
    for k, gn, gm in zip(names, g[tag + '_gnorm'], g[tag + '_gmax']):
        gr = P[k].grad
        if gn < 0:
            assert gr is None or float(gr.abs().max()) == 0.0, k                       # lamb_l / lamb_h: registered, never used
            continue
        if k.endswith('main.3.main.0.bias') and k.startswith('SCM'):
            # the bias in front of InstanceNorm2d: its true gradient is ZERO (the norm removes any per-channel constant); what both sides
            # hold is rounding residue, four orders below the same convolution's weight gradient
            wn = P[k.replace('bias', 'weight')].grad.double().norm().item()
            assert gr.double().norm().item() < 1e-3 * wn and gn < 1e-3 * wn, (k, gr.double().norm().item(), gn, wn)
            continue
        assert abs(gr.double().norm().item() - gn) <= 2e-4 * gn + 1e-6, (k, gr.double().norm().item(), gn)
        assert abs(gr.abs().max().item() - gm) <= 5e-4 * gm + 1e-6, (k, gr.abs().max().item(), gm)
    for key in g.files:
        if key.startswith(tag + '_grad::'):
            k = key.split('::', 1)[1]
            want = torch.from_numpy(g[key])
            assert (P[k].grad - want).abs().max().item() <= 2e-4 * want.abs().max().item() + 1e-6, k
        if key.startswith(tag + '_buf::'):
            k = key.split('::', 1)[1]
            want = torch.from_numpy(g[key])
            assert (bufs[k].double() - want.double()).abs().max().item() <= 1e-5 * max(1.0, want.double().abs().max().item()), k


@pytest.mark.parametrize('tag', ['eval_r2', 'eval_r1_one'])
def test_whole_network_after_eval(tag):
    """the network after .eval() (the trainer's validation pass): BatchNorm2d on its running statistics (tests/golden/sfnet_eval.npz)"""
    g = np.load(GOLDEN.replace('sfnet.npz', 'sfnet_eval.npz'), allow_pickle=False)
    num_res, seed, n, h, w = (int(v) for v in g[tag + '_cfg'])
    P = SO.synth_state(num_res, seed)
    before = {k: v.clone() for k, v in P.items()}
    with torch.no_grad():
        outs = SO.sfnet_forward(P, torch.from_numpy(g[tag + '_x']), num_res, training=False)
    for i, o in enumerate(outs):
        want = torch.from_numpy(g[f'{tag}_out{i}'])
        assert o.shape == want.shape and (o - want).abs().max().item() < 2e-5, (i, (o - want).abs().max().item())
    assert all(torch.equal(P[k], before[k]) for k in P)
    # and the two modes do differ on this state (running statistics != batch statistics): the test would notice a swapped flag
    with torch.no_grad():
        tr = SO.sfnet_forward(P, torch.from_numpy(g[tag + '_x']), num_res, training=True)
    assert n == 1 or (tr[2] - outs[2]).abs().max().item() > 1e-3


@pytest.mark.parametrize('tag,mode', [('test_indoor_r2', 'Indoor'), ('test_outdoor_r1', 'Outdoor')])
def test_inference_network_with_tlsc_pooling(tag, mode):
    """mode = ['test', Indoor | Outdoor]: Gap / Patch_ap / SFconv pool with the TLSC box mean (sfnet_arch_utils.py:11-70, :108-113, :226-229, :247-250)"""
    g = np.load(GOLDEN.replace('sfnet.npz', 'sfnet_eval.npz'), allow_pickle=False)
    num_res, seed, n, h, w = (int(v) for v in g[tag + '_cfg'])
    P = SO.synth_state(num_res, seed)
    with torch.no_grad():
        outs = SO.sfnet_forward(P, torch.from_numpy(g[tag + '_x']), num_res, training=False, tlsc=SO.TLSC_BASE[mode])
        glob = SO.sfnet_forward(P, torch.from_numpy(g[tag + '_x']), num_res, training=False)
    for i, o in enumerate(outs):
        want = torch.from_numpy(g[f'{tag}_out{i}'])
        assert o.shape == want.shape and (o - want).abs().max().item() < 2e-5, (i, (o - want).abs().max().item())
    assert (outs[2] - glob[2]).abs().max().item() > 2e-4             # the global-pool network is a different function


def test_tlsc_box_mean_against_its_integral_image_form():
    """tlsc_avgpool restates AvgPool2d's exact branch (:55-63: cumsum / cumsum, four-corner difference, replicate pad) as a window mean"""
    x = torch.rand(2, 3, 20, 28, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    for base in (246, 210, 128):
        k1, k2 = 20 * base // 256, 28 * base // 256
        s = torch.nn.functional.pad(x.cumsum(-1).cumsum(-2), (1, 0, 1, 0))
        ref = (s[:, :, k1:, k2:] + s[:, :, :-k1, :-k2] - s[:, :, :-k1, k2:] - s[:, :, k1:, :-k2]) / (k1 * k2)
        _h, _w = ref.shape[2:]
        ref = torch.nn.functional.pad(ref, ((28 - _w) // 2, (28 - _w + 1) // 2, (20 - _h) // 2, (20 - _h + 1) // 2), mode='replicate')
        assert (SO.tlsc_avgpool(x, base) - ref).abs().max().item() < 1e-12


@pytest.mark.parametrize('tag', ['dyn3', 'dyn5'])
def test_dynamic_filter(g, tag):
    c, k, n, h, w = (int(v) for v in g[tag + '_cfg'])
    P = {key.split('::', 1)[1]: torch.from_numpy(g[key]).clone() for key in g.files if key.startswith(tag + '_p::')}
    for kk in P:
        if P[kk].dtype == torch.float32 and not SO.is_buffer(kk):
            P[kk].requires_grad_(True)
    x = torch.from_numpy(g[tag + '_x']).clone().requires_grad_(True)
    bufs = {}
    y = SO.dynamic_filter(x, P, '', k, bufs)
    assert (y.detach() - torch.from_numpy(g[tag + '_y'])).abs().max().item() < 1e-5
    (y * torch.from_numpy(g[tag + '_go'])).sum().backward()
    want = torch.from_numpy(g[tag + '_dx'])
    assert (x.grad - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    for key in g.files:
        if key.startswith(tag + '_g::'):
            kk = key.split('::', 1)[1]
            wg = torch.from_numpy(g[key])
            assert (P[kk].grad - wg).abs().max().item() <= 2e-4 * wg.abs().max().item() + 1e-7, kk
        if key.startswith(tag + '_buf::'):
            kk = key.split('::', 1)[1]
            wb = torch.from_numpy(g[key]).double()
            assert (bufs[kk].double() - wb).abs().max().item() <= 1e-5 * max(1.0, wb.abs().max().item()), kk

"""GPU parity of the un-guided classes of the two hot-path files (reference `NAFNet`, `Restormer` with / without dual_pixel_task)
against vectors produced by the reference itself (tests/golden/unguided.npz, make_golden_unguided.py): output, input gradient
(NAFNet), every parameter gradient."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'unguided.npz'))


def _check(net, tag, with_gx):
    names = [str(n) for n in G[tag + '_names']]
    assert [k for k, _ in net.named_parameters()] == names           # registration order = the reference's
    net.load_state_dict({k: torch.from_numpy(G[f'{tag}_p_{k}']) for k in names}, strict=True)
    net = net.cuda()
    x = torch.from_numpy(G[tag + '_x']).cuda().requires_grad_(with_gx)
    out = net(x)
    assert (out.cpu() - torch.from_numpy(G[tag + '_out'])).abs().max().item() < 1e-4
    (out * torch.from_numpy(G[tag + '_go']).cuda()).sum().backward()
    if with_gx:
        gx = torch.from_numpy(G[tag + '_gx'])
        assert (x.grad.cpu() - gx).abs().max().item() < 2e-3 * gx.abs().max().item()
    for i, (k, p) in enumerate(net.named_parameters()):
        want = float(G[tag + '_gnorm'][i])
        assert abs(p.grad.double().norm().item() - want) <= 5e-3 * want + 1e-7, k
        assert abs(p.grad.abs().max().item() - float(G[tag + '_gmax'][i])) <= 5e-3 * float(G[tag + '_gmax'][i]) + 1e-7, k


@pytest.mark.parametrize('math', ['hx2', 'f32'])
def test_unguided_nafnet(math):
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models.archs import define_network
    prev = K.MATH
    K.set_math(math)
    try:
        _check(define_network(dict(type='NAFNet', img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 2], dec_blk_nums=[1, 1, 1])),
               'nafnet', True)
    finally:
        K.set_math(prev)


@pytest.mark.parametrize('math', ['hx2', 'f32'])
def test_unguided_restormer(math):
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models.archs import define_network
    prev = K.MATH
    K.set_math(math)
    try:
        _check(define_network(dict(type='Restormer', inp_channels=3, out_channels=3, dim=8, num_blocks=[1, 2, 1, 1], num_refinement_blocks=1,
                                   heads=[1, 2, 2, 4], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias')), 'restormer', False)
        _check(define_network(dict(type='Restormer', inp_channels=6, out_channels=3, dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
                                   heads=[1, 2, 2, 4], ffn_expansion_factor=2.66, bias=True, LayerNorm_type='BiasFree', dual_pixel_task=True)),
               'restormer_dp', False)
    finally:
        K.set_math(prev)


def test_unguided_restormer_needs_multiples_of_8():
    from textualdegremoval_amd.models.archs import define_network
    net = define_network(dict(type='Restormer', dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 2, 4])).cuda()
    with pytest.raises(ValueError):
        net(torch.rand(1, 3, 36, 40).cuda())

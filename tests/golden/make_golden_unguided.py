"""Generate golden vectors for the UN-GUIDED classes that live in the same reference files as the guided ones -- `NAFNet`
(models/archs/network_nafnet_guided_arch.py:305-386) and `Restormer` (network_restormer_guided_arch.py:396-501, with and without
dual_pixel_task) -- by running the REFERENCE on CPU.

Run in the build container only:   python tests/golden/make_golden_unguided.py
Writes tests/golden/unguided.npz.  Weights are the reference's default init under torch.manual_seed (then every parameter that
starts at zero / one -- beta, gamma, LayerNorm affine, temperature -- gets N(0, 0.1) added so no block is an identity); they are
stored, together with the inputs, because the build's init order must not be assumed.  Only data is saved."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def import_ref(name):
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.' + name)


def run(net, x, tag, d):
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.1 if p.dim() <= 1 or k.endswith(('beta', 'gamma', 'temperature')) else 0)
    x = x.clone().requires_grad_(True)
    out = net(x)
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    d[tag + '_x'], d[tag + '_out'], d[tag + '_go'] = x.detach().numpy(), out.detach().numpy(), go.numpy()
    d[tag + '_gx'] = x.grad.numpy()
    names = [k for k, _ in net.named_parameters()]
    d[tag + '_names'] = np.array(names)
    for k, p in net.named_parameters():
        d[f'{tag}_p_{k}'] = p.detach().numpy()
    d[tag + '_gnorm'] = np.array([p.grad.double().norm().item() if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    d[tag + '_gmax'] = np.array([p.grad.abs().max().item() if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    print(tag, tuple(out.shape), float(out.abs().mean()))


def main():
    d = {}
    naf = import_ref('network_nafnet_guided_arch')
    torch.manual_seed(1)
    net = naf.NAFNet(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 2], dec_blk_nums=[1, 1, 1])
    run(net, torch.rand(2, 3, 44, 60, generator=torch.Generator().manual_seed(2)), 'nafnet', d)       # padded to 48 x 64
    res = import_ref('network_restormer_guided_arch')
    torch.manual_seed(3)
    net = res.Restormer(inp_channels=3, out_channels=3, dim=8, num_blocks=[1, 2, 1, 1], num_refinement_blocks=1, heads=[1, 2, 2, 4],
                        ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', dual_pixel_task=False)
    run(net, torch.rand(2, 3, 32, 64, generator=torch.Generator().manual_seed(4)), 'restormer', d)
    torch.manual_seed(5)
    net = res.Restormer(inp_channels=6, out_channels=3, dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 2, 4],
                        ffn_expansion_factor=2.66, bias=True, LayerNorm_type='BiasFree', dual_pixel_task=True)
    run(net, torch.rand(1, 6, 64, 32, generator=torch.Generator().manual_seed(6)), 'restormer_dp', d)
    np.savez_compressed(os.path.join(HERE, 'unguided.npz'), **d)


if __name__ == '__main__':
    main()

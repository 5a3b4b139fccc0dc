"""Golden vectors of the UN-GUIDED `PromptIR` (models/archs/network_promptir_guided_arch.py:443-590) and `DRSformer`
(network_drsformer_guided_arch.py:586-676), by running the REFERENCE classes on CPU.

Run in the build container only:   python tests/golden/make_golden_unguided2.py
Writes tests/golden/unguided2.npz (data only).  The un-guided classes register a subset of the guided classes' parameters under the
same names, so the weights are the guided oracles' seeded synthesis (oracle.promptir_ref_oracle.synth_params /
oracle.drsformer_ref_oracle.full_synth_params) restricted to the keys the class registers -- regenerated from the seed by the
tests, not stored.  Stored: input, output, cotangent, the gradient norm and maximum of every parameter (-1 where the reference
leaves .grad None), and the parameter-name order of the reference class.
Also recorded: PromptIR(decoder=False) raises in its first forward pass (R4, as the guided class)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'
from oracle import drsformer_ref_oracle as DO  # noqa: E402
from oracle import promptir_ref_oracle as PO  # noqa: E402


def import_ref(name):
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.' + name)


def run(net, P, x, tag, d):
    sd = net.state_dict()
    missing = [k for k in sd if k not in P]
    assert not missing, missing
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    net.load_state_dict({k: P[k] for k in sd})
    out = net(x)
    g = torch.Generator().manual_seed(77)
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    names = [k for k, _ in net.named_parameters()]
    d[tag + '_x'], d[tag + '_out'], d[tag + '_go'], d[tag + '_names'] = x.numpy(), out.detach().numpy(), go.numpy(), np.array(names)
    d[tag + '_gnorm'] = np.array([p.grad.double().norm().item() if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    d[tag + '_gmax'] = np.array([p.grad.abs().max().item() if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    print(tag, tuple(out.shape), float(out.abs().mean()), 'params', len(names), 'without grad', int((d[tag + '_gnorm'] < 0).sum()))


def main():
    d = {}
    pi = import_ref('network_promptir_guided_arch')
    cfg = PO.default_cfg(num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1])
    P = PO.synth_params(cfg, seed=11)
    net = pi.PromptIR(inp_channels=3, out_channels=3, dim=48, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=cfg['heads'],
                      ffn_expansion_factor=cfg['ffn_expansion_factor'], bias=cfg['bias'], LayerNorm_type=cfg['LayerNorm_type'], decoder=True)
    run(net, P, torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(12)), 'promptir', d)    # (latent 8 x 8: rows of 4-pixel multiples, the depthwise stencils' requirement)
    d['promptir_cfg_seed'] = np.array(11)
    try:
        bad = pi.PromptIR(dim=48, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, decoder=False)
        bad(torch.rand(1, 3, 32, 32))
        d['promptir_decoder_false'] = np.array('runs')
    except RuntimeError as e:
        d['promptir_decoder_false'] = np.array('RuntimeError: ' + str(e)[:160])
    print('PromptIR(decoder=False):', d['promptir_decoder_false'])
    dr = import_ref('network_drsformer_guided_arch')
    import functools
    dr.functools = functools                                   # R5: the file never imports functools (recorded by make_golden_drsformer.py)
    cfg = DO.default_cfg(dim=16, nf=16, num_blocks=[1, 1, 1, 1], heads=[1, 2, 2, 4], ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1])
    P = DO.full_synth_params(cfg, seed=13)
    net = dr.DRSformer(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 1, 1, 1], heads=[1, 2, 2, 4],
                       ffn_expansion_factor=cfg['ffn_expansion_factor'], bias=cfg['bias'], LayerNorm_type=cfg['LayerNorm_type'])
    run(net, P, torch.rand(2, 3, 32, 64, generator=torch.Generator().manual_seed(14)), 'drsformer', d)
    d['drsformer_cfg_seed'] = np.array(13)
    np.savez_compressed(os.path.join(HERE, 'unguided2.npz'), **d)
    print('wrote unguided2.npz')


if __name__ == '__main__':
    main()

"""Golden vectors of the UN-GUIDED `SFNet` (models/archs/network_sfnet_guided_arch.py:320-407) and of its `dynamic_filter` / `ResBlock`
operators (models/archs/sfnet_arch_utils.py:120-236), by running the REFERENCE classes on CPU in training mode.

Run in the build container only:   python tests/golden/make_golden_sfnet.py      (--eval-only: only the second file)
Writes tests/golden/sfnet.npz and tests/golden/sfnet_eval.npz (the network after .eval(): outputs only) -- data only.  Weights are oracle.sfnet_oracle.synth_state(num_res, seed) -- regenerated from the seed by
the tests, not stored; the generator first checks that the reference class registers exactly the names / shapes / order the oracle
lists.  Stored per case: input, the three outputs, the cotangents, every parameter's gradient norm and maximum (-1 where the reference
leaves .grad None: the unused lamb_l / lamb_h), a few full gradients, and the BatchNorm buffers after the training-mode forward."""
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
from oracle import sfnet_oracle as SO  # noqa: E402

FULL_GRADS = ('feat_extract.3.main.0.weight', 'Encoder.0.layers.1.dyna.conv.weight', 'Encoder.0.layers.1.dyna.bn.weight',
              'Encoder.0.layers.1.dyna_2.modulate.fcs.0.weight', 'Decoder.1.layers.1.localap.h', 'Decoder.2.layers.0.global_ap.fscale_d',
              'SCM2.main.4.weight', 'Encoder.1.layers.0.conv1.main.0.bias')


def import_ref(name):
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.' + name)


def whole_net(mod, tag, num_res, seed, n, h, w, d):
    net = mod.SFNet(mode=['train', 'Indoor'], num_res=num_res)
    sd = net.state_dict()
    want = SO.state_shapes(num_res)
    assert list(sd) == list(want), 'registration order differs from the oracle list'
    for k in sd:
        assert tuple(sd[k].shape) == tuple(want[k]), (k, sd[k].shape, want[k])
    P = SO.synth_state(num_res, seed)
    net.load_state_dict(P)
    net.train()
    x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(seed + 1))
    outs = net(x)
    g = torch.Generator().manual_seed(seed + 2)
    gos = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * go).sum() for o, go in zip(outs, gos)).backward()
    names = [k for k, _ in net.named_parameters()]
    d[tag + '_x'] = x.numpy()
    for i, (o, go) in enumerate(zip(outs, gos)):
        d[f'{tag}_out{i}'], d[f'{tag}_go{i}'] = o.detach().numpy(), go.numpy()
    d[tag + '_names'] = np.array(names)
    d[tag + '_gnorm'] = np.array([p.grad.double().norm().item() if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    d[tag + '_gmax'] = np.array([p.grad.abs().max().item() if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    par = dict(net.named_parameters())
    for k in FULL_GRADS:
        if k in par and par[k].grad is not None:
            d[f'{tag}_grad::{k}'] = par[k].grad.numpy()
    after = net.state_dict()
    for k in after:
        if SO.is_buffer(k):
            d[f'{tag}_buf::{k}'] = after[k].numpy()
    d[tag + '_cfg'] = np.array([num_res, seed, n, h, w])
    print(tag, [tuple(o.shape) for o in outs], 'params', len(names), 'without grad', int((d[tag + '_gnorm'] < 0).sum()))


def eval_net(mod, tag, num_res, seed, n, h, w, d, mode=('train', 'Indoor')):
    """the same network after .eval() (the trainer's validation pass): BatchNorm2d on its running statistics, no buffer moves.
    mode[0] == 'test': the inference network of the reference -- Gap / Patch_ap / SFconv pool with the TLSC box mean (:108-113, :226-229, :247-250)"""
    net = mod.SFNet(mode=list(mode), num_res=num_res)
    P = SO.synth_state(num_res, seed)
    net.load_state_dict(P)
    net.eval()
    x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(seed + 1))
    with torch.no_grad():
        outs = net(x)
    d[tag + '_x'] = x.numpy()
    for i, o in enumerate(outs):
        d[f'{tag}_out{i}'] = o.numpy()
    after = net.state_dict()
    assert all(torch.equal(after[k], P[k]) for k in after if SO.is_buffer(k)), 'eval moved a buffer'
    d[tag + '_cfg'] = np.array([num_res, seed, n, h, w])
    print(tag, [tuple(o.shape) for o in outs])


def dyn_filter_case(utils, tag, c, k, n, h, w, seed, d):
    m = utils.dynamic_filter(c, ['train', 'Indoor'], kernel_size=k)
    g = torch.Generator().manual_seed(seed)
    sd = m.state_dict()
    P = {}
    for kk, v in sd.items():
        if kk.endswith('num_batches_tracked'):
            P[kk] = torch.tensor(0, dtype=torch.long)
        elif kk.endswith('running_var'):
            P[kk] = torch.rand(v.shape, generator=g) + 0.5
        elif kk.endswith('bn.weight'):
            P[kk] = 1 + 0.2 * torch.randn(v.shape, generator=g)
        else:
            P[kk] = torch.randn(v.shape, generator=g) * (0.3 if v.dim() < 2 else 1.0 / (v.shape[1] ** 0.5))
    m.load_state_dict(P)
    m.train()
    x = torch.randn(n, c, h, w, generator=g).requires_grad_(True)
    y = m(x)
    go = torch.randn(y.shape, generator=g)
    (y * go).sum().backward()
    d[tag + '_x'], d[tag + '_y'], d[tag + '_go'], d[tag + '_dx'] = x.detach().numpy(), y.detach().numpy(), go.numpy(), x.grad.numpy()
    for kk, v in P.items():
        d[f'{tag}_p::{kk}'] = v.numpy()
    for kk, p in m.named_parameters():
        if p.grad is not None:
            d[f'{tag}_g::{kk}'] = p.grad.numpy()
    for kk, v in m.state_dict().items():
        if SO.is_buffer(kk):
            d[f'{tag}_buf::{kk}'] = v.numpy()
    d[tag + '_cfg'] = np.array([c, k, n, h, w])
    print(tag, tuple(y.shape), float(y.abs().mean()))


def main():
    d = {}
    mod = import_ref('network_sfnet_guided_arch')
    utils = sys.modules['models.archs.sfnet_arch_utils']
    whole_net(mod, 'net_r2', 2, 21, 2, 64, 64, d)
    whole_net(mod, 'net_r1_rect', 1, 22, 3, 48, 80, d)              # non-square, odd batch (BatchNorm over 3 samples)
    dyn_filter_case(utils, 'dyn3', 16, 3, 2, 16, 24, 31, d)
    dyn_filter_case(utils, 'dyn5', 32, 5, 3, 12, 8, 32, d)
    if '--eval-only' not in sys.argv:
        np.savez_compressed(os.path.join(HERE, 'sfnet.npz'), **d)
        print('wrote', os.path.join(HERE, 'sfnet.npz'), len(d), 'arrays')
    e = {}
    eval_net(mod, 'eval_r2', 2, 23, 2, 64, 64, e)
    eval_net(mod, 'eval_r1_one', 1, 24, 1, 40, 72, e)               # one image: what a validation loader feeds
    eval_net(mod, 'test_indoor_r2', 2, 25, 1, 64, 96, e, mode=('test', 'Indoor'))
    eval_net(mod, 'test_outdoor_r1', 1, 26, 2, 48, 40, e, mode=('test', 'Outdoor'))
    np.savez_compressed(os.path.join(HERE, 'sfnet_eval.npz'), **e)
    print('wrote', os.path.join(HERE, 'sfnet_eval.npz'), len(e), 'arrays')


if __name__ == '__main__':
    main()

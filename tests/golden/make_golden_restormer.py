"""Generate Restormer-ref golden vectors by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference, which never travels):
    python tests/golden/make_golden_restormer.py
Writes tests/golden/restormer_*.npz (data only: inputs and weights are regenerated from seeds by
oracle.restormer_ref_oracle.synth_params / oracle.nafnet_ref_oracle.synth_pair, outputs are stored).

Reference defect worked around here: RestormerRefFusion.forward indexes the
encoder pyramid one slot off; Encoder.forward is wrapped to return [None, L1, L2, L3, L4] -- the only
assignment under which the reference code runs.  Nothing from the reference is copied: it is imported,
executed, and only its numeric outputs are saved.
"""
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

from oracle import nafnet_ref_oracle as NO  # noqa: E402
from oracle import restormer_ref_oracle as RO  # noqa: E402


def import_ref_arch():
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.network_restormer_guided_arch')


def sample(t, n=32):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def build_ref_net(arch, cfg, P):
    net = arch.RestormerRefFusion(
        inp_channels=cfg['inp_channels'], out_channels=cfg['out_channels'], dim=cfg['dim'],
        num_blocks=cfg['num_blocks'], num_refinement_blocks=cfg['num_refinement_blocks'], heads=cfg['heads'],
        ffn_expansion_factor=cfg['ffn_expansion_factor'], bias=cfg['bias'], LayerNorm_type=cfg['LayerNorm_type'],
        nf=cfg['nf'], ext_n_blocks=cfg['ext_n_blocks'], reffusion_n_blocks=cfg['reffusion_n_blocks'],
        lr_block_size=cfg['lr_block_size'], ref_down_block_size=cfg['ref_down_block_size'],
        dilations=cfg['dilations'], psize=cfg['psize'])
    sd = net.state_dict()
    assert list(sd.keys()) == list(P.keys()), 'registration order mismatch'
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    net.load_state_dict(P)
    enc_forward = net.masa_enc.forward
    net.masa_enc.forward = lambda x: [None] + list(enc_forward(x))          # R1
    return net


def whole_net_case(arch, name, cfg, B, H, W, seed):
    P = RO.synth_params(cfg, seed=seed)
    net = build_ref_net(arch, cfg, P)
    lq, gt, ref = NO.synth_pair(B, H, W, seed=4321 + seed)
    rec = {}
    orig_search, orig_search_org, orig_transfer = net.search, net.search_org, net.transfer

    def search(*a, **k):
        r = orig_search(*a, **k); rec['index'] = r[1].detach(); return r

    def search_org(*a, **k):
        r = orig_search_org(*a, **k); rec['soft'] = r[0].detach(); rec['index_all'] = r[1].detach(); return r
    warps = []

    def transfer(*a, **k):
        r = orig_transfer(*a, **k); warps.append(r); return r
    net.search, net.search_org, net.transfer = search, search_org, transfer
    out = net(lq, ref)
    loss = (out - gt).abs().mean()
    loss.backward()
    d = dict(out=out.detach().numpy(), loss=np.float64(loss.item()), index=rec['index'].numpy(),
             index_all=rec['index_all'][..., 0].numpy(), soft_att=rec['soft'][..., 0].numpy(),
             cfg_B=B, cfg_H=H, cfg_W=W, seed=seed, ln_type=cfg['LayerNorm_type'], bias=cfg['bias'])
    for i, wv in enumerate(warps):           # order x1,x2,x4,x8 (coarse->fine)
        d[f'warp{i}_stats'] = stats(wv)
        d[f'warp{i}_sample'] = sample(wv, 64)
    with torch.no_grad():
        _, aux = RO.restormer_ref_forward(P, cfg, lq, ref, return_aux=True)
    assert torch.equal(aux['index_all'], rec['index_all'][..., 0])
    t2 = aux['corr_fine'].topk(2, dim=2).values
    d['fine_gap'] = (t2[..., 0] - t2[..., 1]).numpy()
    c2 = aux['corr_sum'].topk(2, dim=2).values
    d['coarse_gap'] = (c2[..., 0] - c2[..., 1]).numpy()
    names = list(P.keys())
    gnorm = np.zeros(len(names)); gsum = np.zeros(len(names)); gsample = np.zeros((len(names), 8), dtype=np.float32)
    has_grad = np.zeros(len(names), dtype=bool)
    for i, (k, p) in enumerate(net.named_parameters()):
        assert k == names[i]
        has_grad[i] = p.grad is not None
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        gnorm[i] = g.double().norm().item(); gsum[i] = g.double().sum().item()
        s = sample(g, 8); gsample[i, :len(s)] = s
    d['grad_norm'] = gnorm; d['grad_sum'] = gsum; d['grad_sample'] = gsample; d['has_grad'] = has_grad
    d['total_grad_norm'] = np.float64(np.sqrt((gnorm ** 2).sum()))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print(name, 'loss', loss.item(), 'gnorm', d['total_grad_norm'], 'idx', rec['index'].flatten()[:8].tolist(),
          'min fine gap', d['fine_gap'].min(), 'min coarse gap', d['coarse_gap'].min(),
          'no-grad params', [names[i] for i in range(len(names)) if not has_grad[i]][:4])


def per_op_cases(arch):
    d = {}
    g = torch.Generator().manual_seed(17)

    def fill(mod, base):
        with torch.no_grad():
            for i, (k, p) in enumerate(mod.named_parameters()):
                gg = torch.Generator().manual_seed(base + i)
                if k.endswith('temperature'):
                    p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=gg))
                elif 'norm' in k or k.startswith('body'):
                    p.copy_(torch.randn(p.shape, generator=gg) * 0.2 + (1.0 if k.endswith('weight') else 0.0))
                elif k == 'alpha':
                    p.copy_(torch.randn(p.shape, generator=gg) * 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=gg) * 0.25)

    def run(tag, mod, x, out_slice=None):
        go = None
        y = mod(x)
        go = torch.randn(y.shape, generator=g)
        y.backward(go)
        d[tag + '_x'] = x.detach().numpy(); d[tag + '_go'] = go.numpy()
        d[tag + '_y'] = y.detach().numpy(); d[tag + '_gx'] = x.grad.numpy()
        d[tag + '_names'] = np.array([k for k, _ in mod.named_parameters()])
        for k, p in mod.named_parameters():
            d[f'{tag}_p_{k}'] = p.detach().numpy(); d[f'{tag}_g_{k}'] = p.grad.numpy()

    # a13 both LayerNorm flavours, C=12 @ 9x10
    for kind in ('BiasFree', 'WithBias'):
        ln = arch.LayerNorm(12, kind); fill(ln, 10)
        run('ln_' + kind, ln, (torch.randn(2, 12, 9, 10, generator=g) + 0.5).requires_grad_())
    # a14 GDFN dim 12 (hidden 31), with and without bias
    for b in (False, True):
        ff = arch.FeedForward(12, 2.66, b); fill(ff, 40)
        run(f'gdfn_b{int(b)}', ff, torch.randn(2, 12, 8, 12, generator=g).requires_grad_())
    # a15 MDTA dim 16, heads 2, with and without bias
    for b in (False, True):
        at = arch.Attention(16, 2, b); fill(at, 70)
        run(f'mdta_b{int(b)}', at, torch.randn(2, 16, 8, 12, generator=g).requires_grad_())
    # a16 TransformerBlock / TransformerResFusionBlock, dim 16 heads 4
    for kind in ('BiasFree', 'WithBias'):
        tb = arch.TransformerBlock(16, 4, 2.66, False, kind); fill(tb, 100)
        run('tblock_' + kind, tb, torch.randn(2, 16, 8, 8, generator=g).requires_grad_())
    fb = arch.TransformerResFusionBlock(16, 2, 2.66, False, 'WithBias'); fill(fb, 130)
    run('fblock', fb, torch.randn(2, 16, 8, 8, generator=g).requires_grad_())
    # a17 Downsample / Upsample
    dn = arch.Downsample(8); fill(dn, 160)
    run('down', dn, torch.randn(2, 8, 8, 12, generator=g).requires_grad_())
    up = arch.Upsample(8); fill(up, 170)
    run('up', up, torch.randn(2, 8, 6, 8, generator=g).requires_grad_())
    np.savez_compressed(os.path.join(HERE, 'restormer_per_op.npz'), **d)
    print('restormer per_op done')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    arch = import_ref_arch()
    per_op_cases(arch)
    whole_net_case(arch, 'restormer_d8_128', RO.default_cfg(), 1, 128, 128, seed=1)
    whole_net_case(arch, 'restormer_d8_128_biasfree_b2',
                   RO.default_cfg(LayerNorm_type='BiasFree', num_blocks=[1, 2, 1, 1]), 2, 128, 128, seed=2)
    whole_net_case(arch, 'restormer_d8_64_wrap_bias', RO.default_cfg(bias=True), 1, 64, 64, seed=3)
    whole_net_case(arch, 'restormer_d16_120x100_pad', RO.default_cfg(dim=16, nf=16), 1, 120, 100, seed=4)

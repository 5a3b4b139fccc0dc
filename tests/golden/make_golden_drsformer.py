"""Generate DRSformer-ref golden vectors by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference, which never travels):
    python tests/golden/make_golden_drsformer.py
Writes tests/golden/drsformer_*.npz (data only: inputs and weights are regenerated from seeds by
oracle.drsformer_ref_oracle.synth_params / oracle.nafnet_ref_oracle.synth_pair, outputs are stored).

Class under test: DRSformer200L_SPA_RefFusion (models/archs/network_drsformer_guided_arch_200L_SPA.py, the network of
007_drsformer_image_deraining_rain200l.yml).  Reference defects worked around (oracle/drsformer_ref_oracle.py docstring):
R5 -- the file uses functools without importing it (NameError at construction): `functools` is injected into the module
namespace; R1 -- Encoder.forward is wrapped to return [None, L1, L2, L3, L4].  Both failures are recorded in
drsformer_defects.npz.  Also written: per-op vectors of the reference's Attention (top-k sparse attention) and FeedForward
(mixed-scale) classes.  Nothing from the reference is copied: it is imported, executed, and only its numeric outputs are saved.
"""
import functools
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
from oracle import drsformer_ref_oracle as DO  # noqa: E402


def import_ref_arch():
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.network_drsformer_guided_arch_200L_SPA')


def sample(t, n=32):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def build_ref_net(arch, cfg, P=None):
    net = arch.DRSformer200L_SPA_RefFusion(
        inp_channels=cfg['inp_channels'], out_channels=cfg['out_channels'], dim=cfg['dim'],
        num_blocks=cfg['num_blocks'], heads=cfg['heads'],
        ffn_expansion_factor=cfg['ffn_expansion_factor'], bias=cfg['bias'], LayerNorm_type=cfg['LayerNorm_type'],
        nf=cfg['nf'], ext_n_blocks=cfg['ext_n_blocks'], reffusion_n_blocks=cfg['reffusion_n_blocks'],
        lr_block_size=cfg['lr_block_size'], ref_down_block_size=cfg['ref_down_block_size'],
        dilations=cfg['dilations'], psize=cfg['psize'])
    if P is not None:
        sd = net.state_dict()
        assert list(sd.keys()) == list(P.keys()), 'registration order mismatch'
        for k in sd:
            assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
        net.load_state_dict(P)
    enc_forward = net.masa_enc.forward
    net.masa_enc.forward = lambda x: [None] + list(enc_forward(x))          # R1
    return net


def whole_net_case(arch, name, cfg, B, H, W, seed):
    P = DO.synth_params(cfg, seed=seed)
    net = build_ref_net(arch, cfg, P)
    lq, gt, ref = NO.synth_pair(B, H, W, seed=8765 + seed)
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
    with torch.no_grad():
        _, aux = DO.drsformer_ref_forward(P, cfg, lq, ref, return_aux=True)
    assert torch.equal(aux['index_all'], rec['index_all'][..., 0])
    t2 = aux['corr_fine'].topk(2, dim=2).values
    d['fine_gap'] = (t2[..., 0] - t2[..., 1]).numpy()
    c2 = aux['corr_sum'].topk(2, dim=2).values
    d['coarse_gap'] = (c2[..., 0] - c2[..., 1]).numpy()
    names = list(P.keys())
    grads = {k: p.grad for k, p in net.named_parameters()}
    gnorm = np.zeros(len(names)); gsum = np.zeros(len(names)); gsample = np.zeros((len(names), 8), dtype=np.float32)
    has_grad = np.zeros(len(names), dtype=bool)
    for i, k in enumerate(names):
        g = grads[k]
        has_grad[i] = g is not None
        g = g if g is not None else torch.zeros_like(P[k])
        gnorm[i] = g.double().norm().item(); gsum[i] = g.double().sum().item()
        s = sample(g, 8); gsample[i, :len(s)] = s
    d['names'] = np.array(names)
    d['grad_norm'] = gnorm; d['grad_sum'] = gsum; d['grad_sample'] = gsample; d['has_grad'] = has_grad
    d['total_grad_norm'] = np.float64(np.sqrt((gnorm ** 2).sum()))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print(name, 'loss', loss.item(), 'gnorm', d['total_grad_norm'], 'idx', rec['index'].flatten()[:8].tolist(),
          'min fine gap', d['fine_gap'].min(), 'min coarse gap', d['coarse_gap'].min(),
          'no-grad params', sorted({names[i].split('.')[0] for i in range(len(names)) if not has_grad[i]}))


def import_full_arch():
    return importlib.import_module('models.archs.network_drsformer_guided_arch')


def full_net_case(arch_full, name, cfg, B, H, W, seed):
    """DRSformerRefFusion (with the MEFC sub-networks, network_drsformer_guided_arch.py:679-1123); only R1 is wrapped."""
    P = DO.full_synth_params(cfg, seed=seed)
    net = arch_full.DRSformerRefFusion(
        inp_channels=cfg['inp_channels'], out_channels=cfg['out_channels'], dim=cfg['dim'], num_blocks=cfg['num_blocks'],
        heads=cfg['heads'], ffn_expansion_factor=cfg['ffn_expansion_factor'], bias=cfg['bias'], LayerNorm_type=cfg['LayerNorm_type'],
        nf=cfg['nf'], ext_n_blocks=cfg['ext_n_blocks'], reffusion_n_blocks=cfg['reffusion_n_blocks'],
        lr_block_size=cfg['lr_block_size'], ref_down_block_size=cfg['ref_down_block_size'], dilations=cfg['dilations'], psize=cfg['psize'])
    sd = net.state_dict()
    assert list(sd.keys()) == list(P.keys()), 'registration order mismatch'
    net.load_state_dict(P)
    enc_forward = net.masa_enc.forward
    net.masa_enc.forward = lambda x: [None] + list(enc_forward(x))          # R1
    lq, gt, ref = NO.synth_pair(B, H, W, seed=8765 + seed)
    out = net(lq, ref)
    loss = (out - gt).abs().mean()
    loss.backward()
    names = list(P.keys())
    grads = {k: p.grad for k, p in net.named_parameters()}
    gnorm = np.zeros(len(names)); gsample = np.zeros((len(names), 8), dtype=np.float32); has_grad = np.zeros(len(names), dtype=bool)
    for i, k in enumerate(names):
        g = grads[k]
        has_grad[i] = g is not None
        g = g if g is not None else torch.zeros_like(P[k])
        gnorm[i] = g.double().norm().item()
        sm = sample(g, 8); gsample[i, :len(sm)] = sm
    with torch.no_grad():
        _, aux = DO.drsformer_full_forward(P, cfg, lq, ref, return_aux=True)
    t2 = aux['corr_fine'].topk(2, dim=2).values
    np.savez_compressed(os.path.join(HERE, name + '.npz'), out=out.detach().numpy(), loss=np.float64(loss.item()),
                        index_all=aux['index_all'].numpy(), fine_gap=(t2[..., 0] - t2[..., 1]).numpy(), names=np.array(names),
                        grad_norm=gnorm, grad_sample=gsample, has_grad=has_grad, cfg_B=B, cfg_H=H, cfg_W=W, seed=seed,
                        total_grad_norm=np.float64(np.sqrt((gnorm ** 2).sum())))
    print(name, 'loss', loss.item(), 'gnorm', np.sqrt((gnorm ** 2).sum()), 'all params have grads:', bool(has_grad.all()))


def mefc_case(arch_full):
    """`subnet` alone (:522-548): OALayer gating + 4 weighted-operation steps over the 8 candidate operations"""
    d = {}
    g = torch.Generator().manual_seed(31)
    for tag, (C, H, W) in {'mefc_a': (8, 12, 16), 'mefc_b': (12, 9, 10)}.items():
        sub = arch_full.subnet(C)
        with torch.no_grad():
            for i, (k, p) in enumerate(sub.named_parameters()):
                gg = torch.Generator().manual_seed(900 + i)
                p.copy_(torch.randn(p.shape, generator=gg) * (0.5 if 'ca_fc' in k else 0.35))
        x = torch.randn(2, C, H, W, generator=g).requires_grad_()
        y = sub(x)
        go = torch.randn(y.shape, generator=g)
        y.backward(go)
        d[tag + '_x'] = x.detach().numpy(); d[tag + '_go'] = go.numpy(); d[tag + '_y'] = y.detach().numpy(); d[tag + '_gx'] = x.grad.numpy()
        d[tag + '_names'] = np.array([k for k, _ in sub.named_parameters()])
        for k, p in sub.named_parameters():
            d[f'{tag}_p_{k}'] = p.detach().numpy(); d[f'{tag}_g_{k}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'drsformer_mefc.npz'), **d)
    print('mefc cases done')


def per_op_cases(arch):
    """the two block classes that differ from Restormer-ref: Attention (TKSA, :257-328) and FeedForward (MSFN, :213-253)"""
    d = {}
    g = torch.Generator().manual_seed(29)

    def fill(mod, base):
        with torch.no_grad():
            for i, (k, p) in enumerate(mod.named_parameters()):
                gg = torch.Generator().manual_seed(base + i)
                if k == 'temperature':
                    p.copy_(4.0 + 0.8 * torch.randn(p.shape, generator=gg))
                elif k.startswith('attn'):
                    p.copy_(0.2 + 0.1 * torch.randn(p.shape, generator=gg))
                else:
                    p.copy_(torch.randn(p.shape, generator=gg) * 0.25)

    def run(tag, mod, x):
        y = mod(x)
        go = torch.randn(y.shape, generator=g)
        y.backward(go)
        d[tag + '_x'] = x.detach().numpy(); d[tag + '_go'] = go.numpy()
        d[tag + '_y'] = y.detach().numpy(); d[tag + '_gx'] = x.grad.numpy()
        d[tag + '_names'] = np.array([k for k, _ in mod.named_parameters()])
        for k, p in mod.named_parameters():
            d[f'{tag}_p_{k}'] = p.detach().numpy(); d[f'{tag}_g_{k}'] = p.grad.numpy()

    for tag, (dim, heads, H, W) in {'tksa_a': (16, 2, 8, 12), 'tksa_b': (48, 1, 8, 8), 'tksa_c': (24, 4, 12, 8)}.items():
        at = arch.Attention(dim, heads, False); fill(at, 70)
        run(tag, at, torch.randn(2, dim, H, W, generator=g).requires_grad_())
    for tag, (dim, H, W) in {'msfn_a': (12, 8, 12), 'msfn_b': (8, 12, 8)}.items():      # hidden 31 (odd) / 21 (odd)
        ff = arch.FeedForward(dim, 2.66, False); fill(ff, 40)
        run(tag, ff, torch.randn(2, dim, H, W, generator=g).requires_grad_())
    np.savez_compressed(os.path.join(HERE, 'drsformer_per_op.npz'), **d)
    print('drsformer per_op done')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    arch = import_ref_arch()
    cfg = DO.default_cfg()
    defects = {}
    try:
        build_ref_net(arch, cfg, None)
        defects['R5_no_functools'] = ''
    except NameError as e:
        defects['R5_no_functools'] = str(e)[:200]
    arch.functools = functools                       # R5
    net = build_ref_net(arch, cfg, None)
    net.masa_enc.forward = net.masa_enc.__class__.forward.__get__(net.masa_enc)     # undo the R1 wrap: the network as written
    lq, gt, ref = NO.synth_pair(1, 64, 64, seed=1)
    try:
        with torch.no_grad():
            net(lq, ref)
        defects['R1_pyramid_index'] = ''
    except IndexError as e:
        defects['R1_pyramid_index'] = str(e)[:200]
    print('defects:', defects)
    np.savez_compressed(os.path.join(HERE, 'drsformer_defects.npz'), **{k: np.array(v) for k, v in defects.items()})
    per_op_cases(arch)
    whole_net_case(arch, 'drsformer_d8_64', cfg, 1, 64, 64, seed=1)
    whole_net_case(arch, 'drsformer_d8_128_b2_biasfree', DO.default_cfg(LayerNorm_type='BiasFree', num_blocks=[1, 1, 2, 1]), 2, 128, 128, seed=2)
    whole_net_case(arch, 'drsformer_d16_100x72_pad', DO.default_cfg(dim=16, nf=16, bias=True), 1, 100, 72, seed=3)
    full = import_full_arch()
    mefc_case(full)
    full_net_case(full, 'drsformer_full_d8_64', DO.default_cfg(), 1, 64, 64, seed=4)
    full_net_case(full, 'drsformer_full_d8_128_b2', DO.default_cfg(LayerNorm_type='BiasFree'), 2, 128, 128, seed=5)

"""GPU parity of the PromptIR-ref path (first "next" architecture) through the C ABI: PromptGenBlock against
golden vectors of the reference's own class, the whole network forward + backward against reference-generated goldens and
the oracle, the nn.Module mirror, and the train step (eager, capture, replay) against the oracle trainer.
Path target: 1e-4 max-abs on fp32 outputs (north_star); integer indices exact."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as NO
from oracle import promptir_ref_oracle as PO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', params=['bx3', 'f32', 'hx2'])
def PE(request):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels, promptir_engine
    prev = kernels.MATH
    kernels.set_math(request.param)
    yield promptir_engine
    kernels.set_math(prev)


def gold(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def dev(a):
    return T(a).cuda().contiguous()


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


@pytest.mark.parametrize('tag', ['same', 'down', 'up'])
def test_prompt_gen_block_vs_reference_golden(PE, tag):
    """:417-441 with the identity, a 2x down- and an anisotropic up-resize of the prompt components."""
    from textualdegremoval_amd import kernels as K
    g = gold('promptir_prompt_block')
    names = ('prompt_param', 'linear_layer.weight', 'linear_layer.bias', 'conv3x3.weight')
    P = {'p.' + k: dev(g[f'{tag}_p_{k}']) for k in names}
    x = dev(g[tag + '_x'])
    y, saved = PE.prompt_fwd(x, P, 'p.')
    assert maxdiff(y, T(g[tag + '_y'])) < 2e-5
    G = {}
    demb, inv = PE.prompt_bwd(dev(g[tag + '_go']), P, 'p.', saved, G)
    dx = K.plane_add_(torch.zeros_like(x), demb, inv)
    assert maxdiff(dx, T(g[tag + '_gx'])) < 2e-5
    for k in names:
        ref = T(g[f'{tag}_g_{k}'])
        assert maxdiff(G['p.' + k].view_as(ref), ref) < 1e-4 * max(1.0, ref.abs().max().item()), k


def test_bilinear_adjoint_is_the_transpose(PE):
    """<resize(a), b> == <a, resize_bwd(b)> for up- and down-scaling, odd sizes included."""
    from textualdegremoval_amd import kernels as K
    gen = torch.Generator().manual_seed(3)
    for (hs, ws, hd, wd) in [(16, 16, 8, 8), (8, 8, 20, 12), (7, 9, 13, 5), (12, 12, 12, 12), (5, 6, 15, 18)]:
        a = torch.randn(3, 4, hs, ws, generator=gen).cuda()
        b = torch.randn(3, 4, hd, wd, generator=gen).cuda()
        lhs = (K.resize_bilinear(a, hd, wd).double() * b.double()).sum().item()
        rhs = (a.double() * K.resize_bilinear_bwd(b, hs, ws).double()).sum().item()
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (hs, ws, hd, wd, lhs, rhs)
        ref = torch.nn.functional.interpolate(a.cpu(), (hd, wd), mode='bilinear')
        assert maxdiff(K.resize_bilinear(a, hd, wd), ref) < 1e-5


CASES = [('promptir_d48_64', dict()),
         ('promptir_d48_128_b2_biasfree', dict(LayerNorm_type='BiasFree', num_blocks=[1, 1, 2, 1])),
         ('promptir_d48_100x72_pad', dict(bias=True))]


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_vs_reference_golden(PE, name, kw):
    g = gold(name)
    cfg = PO.default_cfg(**kw)
    seed = int(g['seed'])
    P = PO.synth_params(cfg, seed=seed)
    names = [k for k in P if not k.startswith(PO.UNUSED)]
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=8765 + seed)
    Pc = {k: P[k].cuda().contiguous() for k in names}
    out, saved = PE.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
    sv_masa = saved[6]
    index, index_all = sv_masa[4], sv_masa[7]
    assert np.array_equal(index.cpu().numpy().reshape(g['index'].shape[0], -1), g['index'].reshape(g['index'].shape[0], -1))
    # exact-tie positions (zero-padded borders: identical candidate patches, fine_gap == 0) may resolve to either index
    ia = index_all.cpu().numpy().reshape(g['index_all'].shape)
    decided = g['fine_gap'].reshape(g['index_all'].shape) > 1e-6
    assert np.array_equal(ia[decided], g['index_all'][decided])
    assert (ia != g['index_all']).mean() < 0.1
    assert maxdiff(out, T(g['out'])) < 1e-4
    from textualdegremoval_amd import kernels as K
    loss, dpred = K.l1_loss(out.contiguous(), gt.cuda(), 1.0)
    assert abs(loss.item() - float(g['loss'])) < 2e-6
    G = PE.net_bwd(dpred, Pc, cfg, saved)
    assert set(G.keys()) == set(names)
    gnames = [str(n) for n in g['names']]
    for k in names:
        i = gnames.index(k)
        assert bool(g['has_grad'][i]), k
        gn = G[k].double().norm().item()
        assert abs(gn - g['grad_norm'][i]) <= 5e-3 * g['grad_norm'][i] + 1e-5, (k, gn, g['grad_norm'][i])
        s = G[k].detach().reshape(-1)
        step = max(1, s.numel() // 8)
        smp = s[::step][:8].cpu().numpy()
        assert np.abs(smp - g['grad_sample'][i, :len(smp)]).max() < 2e-4 * max(1.0, g['grad_norm'][i]), k
    assert all(not bool(h) for n, h in zip(gnames, g['has_grad']) if n.startswith(PO.UNUSED))


def test_module_autograd_matches_oracle(PE):
    """nn.Module drop-in (define_network) forward + autograd backward == oracle autograd; unused tensors keep grad None."""
    from textualdegremoval_amd.models.archs import define_network
    cfg = PO.default_cfg(LayerNorm_type='BiasFree')
    P = PO.synth_params(cfg, seed=7)
    net = define_network(dict(type='PromptIRRefFusion', **cfg)).cuda()
    assert list(net.state_dict().keys()) == list(P.keys())
    net.load_state_dict(P, strict=True)
    lq, gt, ref = NO.synth_pair(1, 128, 128, seed=77)
    out = net(lq.cuda(), ref.cuda())
    (out - gt.cuda()).abs().mean().backward()
    oo, _, grads = PO.loss_and_grads(P, cfg, lq, ref, gt)
    assert maxdiff(out, oo) < 1e-4
    for k, p in net.named_parameters():
        if k.startswith(PO.UNUSED):
            assert p.grad is None
            continue
        ref_g = grads[k]
        assert maxdiff(p.grad, ref_g) < 3e-4 * max(1e-3, ref_g.abs().max().item()) + 1e-7, k
    with pytest.raises(ValueError):
        define_network(dict(type='PromptIRRefFusion', **dict(cfg, decoder=False)))


def test_train_step_matches_oracle_trainer():
    """RefGuidedImageCleanModel.optimize_parameters with a PromptIR-ref net_g: eager, eager, graph capture, replay; the
    never-used tensors are left untouched (no decay), as torch.optim.AdamW leaves tensors whose grad is None."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd.models import create_model
    cfg = PO.default_cfg()
    opt = {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 1, 'dist': False, 'is_train': True,
        'network_g': dict(type='PromptIRRefFusion', **cfg), 'path': {},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                  'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'val': {}, 'scale': 1,
    }
    model = create_model(opt)
    P = PO.synth_params(cfg, seed=5)
    model.net_g.load_state_dict(P, strict=True)
    tr = NO.OracleTrainer(P, cfg, forward_fn=PO.promptir_ref_forward)
    lq, gt, ref = NO.synth_pair(1, 64, 64, seed=55)
    periods, rw, em = [30, 70], [1, 1], [3e-4, 1e-6]
    for it in range(1, 5):
        t = it - 1
        tr.set_lrs(NO.cosine_restart_cyclic_lr(t, 2e-4, periods, rw, em), NO.cosine_restart_cyclic_lr(t, 1e-4, periods, rw, em))
        loss_o, _, out_o = tr.step(lq, gt, ref)
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref})
        model.optimize_parameters(it)
        assert abs(model.get_current_log()['l_pix'] - loss_o) < 3e-6, (it, model.get_current_log()['l_pix'], loss_o)
    sd = model.net_g.state_dict()
    for k, v in tr.P.items():
        assert maxdiff(sd[k], v) < 2e-5, k
        if k.startswith(PO.UNUSED):
            assert torch.equal(sd[k].cpu(), P[k]), k
    assert maxdiff(model.output, out_o) < 1e-4

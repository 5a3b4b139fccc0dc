"""GPU parity of the Restormer-ref path through the C ABI: kernels and blocks against
golden vectors produced by the reference's own classes (tests/golden/restormer_*.npz), the whole network
forward+backward against those goldens and the oracle, and the train step against the oracle trainer.
Path target: 1e-4 max-abs on fp32 outputs (north_star); integer indices exact."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as NO
from oracle import restormer_ref_oracle as RO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', params=['bx3', 'f32', 'hx2'])
def R(request):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels, restormer_engine
    prev = kernels.MATH
    kernels.set_math(request.param)
    yield restormer_engine
    kernels.set_math(prev)


@pytest.fixture(scope='module')
def K():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels
    return kernels


def gold(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def dev(a):
    return T(a).cuda().contiguous()


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def gparams(g, tag):
    return {str(k): dev(g[f'{tag}_p_{k}']) for k in g[tag + '_names']}


def check_grads(g, tag, G, tol=1e-4):
    for k in g[tag + '_names']:
        ref = T(g[f'{tag}_g_{k}'])
        assert maxdiff(G[str(k)].view_as(ref), ref) < tol * max(1.0, ref.abs().max().item()), k


# ------------------------------------------------------------------ a13 LayerNorm flavours
@pytest.mark.parametrize('kind', ['BiasFree', 'WithBias'])
def test_layernorm_variants_vs_reference_golden(K, kind):
    g = gold('restormer_per_op')
    tag = 'ln_' + kind
    center = kind == 'WithBias'
    w = dev(g[tag + '_p_body.weight'])
    b = dev(g[tag + '_p_body.bias']) if center else None
    x = dev(g[tag + '_x'])
    y, mu, rs = K.layernorm2d_fwd(x, w, b, 1e-5, center=center)
    assert maxdiff(y, T(g[tag + '_y'])) < 2e-5
    gx, gw, gb = K.layernorm2d_bwd(dev(g[tag + '_go']), x, mu, rs, w, center=center)
    assert maxdiff(gx, T(g[tag + '_gx'])) < 2e-5
    assert maxdiff(gw, T(g[tag + '_g_body.weight'])) < 1e-4
    if center:
        assert maxdiff(gb, T(g[tag + '_g_body.bias'])) < 1e-4


@pytest.mark.parametrize('C,HW', [(48, (16, 20)), (96, (8, 12)), (192, (8, 8)), (768, (4, 8))])
def test_layernorm_biasfree_all_kernel_paths_vs_oracle(K, C, HW):
    """every register/generic LN kernel variant with center=0 (C picks the variant)."""
    gen = torch.Generator().manual_seed(C)
    x = (torch.randn(2, C, *HW, generator=gen) + 0.3).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(C, generator=gen)).requires_grad_(True)
    go = torch.randn(2, C, *HW, generator=gen)
    y = RO.layernorm(x, {'n.body.weight': w}, 'n.', 'BiasFree')
    y.backward(go)
    yk, mu, rs = K.layernorm2d_fwd(x.detach().cuda(), w.detach().cuda(), None, 1e-5, center=False)
    assert maxdiff(yk, y) < 2e-5
    add = torch.randn(2, C, *HW, generator=gen)
    gx, gw, _ = K.layernorm2d_bwd(go.cuda(), x.detach().cuda(), mu, rs, w.detach().cuda(), add=add.cuda(), center=False)
    assert maxdiff(gx, x.grad + add) < 3e-5
    assert maxdiff(gw, w.grad) < 2e-4 * max(1.0, w.grad.abs().max().item())


# ------------------------------------------------------------------ depthwise stencils (a14 gate, a15 qkv_dwconv)
@pytest.mark.parametrize('bias', [False, True])
def test_dwgelu_gate_vs_torch(K, bias):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(3)
    N, h, H, W = 2, 21, 12, 16
    t = torch.randn(N, 2 * h, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(2 * h, 1, 3, 3, generator=gen) * 0.3).requires_grad_(True)
    b = (torch.randn(2 * h, generator=gen) * 0.2).requires_grad_(True) if bias else None
    u = F.conv2d(t, w, b, padding=1, groups=2 * h)
    x1, x2 = u.chunk(2, dim=1)
    g = F.gelu(x1) * x2
    go = torch.randn(N, h, H, W, generator=gen)
    g.backward(go)
    gk = K.dwgelu_fwd(t.detach().cuda(), w.detach().cuda(), b.detach().cuda() if bias else None)
    assert maxdiff(gk, g) < 1e-5
    dt, dw, db = K.dwgelu_bwd(go.cuda(), t.detach().cuda(), w.detach().cuda(), b.detach().cuda() if bias else None)
    assert maxdiff(dt, t.grad) < 2e-5
    assert maxdiff(dw, w.grad) < 1e-4 * max(1.0, w.grad.abs().max().item())
    if bias:
        assert maxdiff(db, b.grad) < 1e-4 * max(1.0, b.grad.abs().max().item())


@pytest.mark.parametrize('bias', [False, True])
def test_plain_depthwise_vs_torch(K, bias):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(4)
    N, Pn, H, W = 2, 24, 10, 8
    t = torch.randn(N, Pn, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(Pn, 1, 3, 3, generator=gen) * 0.3).requires_grad_(True)
    b = (torch.randn(Pn, generator=gen) * 0.2).requires_grad_(True) if bias else None
    o = F.conv2d(t, w, b, padding=1, groups=Pn)
    go = torch.randn(N, Pn, H, W, generator=gen)
    o.backward(go)
    ok = K.dwconv_fwd(t.detach().cuda(), w.detach().cuda(), b.detach().cuda() if bias else None)
    assert maxdiff(ok, o) < 1e-5
    dt, dw, db = K.dwconv_bwd(go.cuda(), t.detach().cuda(), w.detach().cuda(), want_db=bias)
    assert maxdiff(dt, t.grad) < 1e-5
    assert maxdiff(dw, w.grad) < 1e-4 * max(1.0, w.grad.abs().max().item())
    if bias:
        assert maxdiff(db, b.grad) < 1e-4 * max(1.0, b.grad.abs().max().item())


def test_pixel_shuffle_and_glue(K):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(2, 12, 5, 8, generator=gen)
    assert torch.equal(K.pixel_shuffle2(x.cuda()).cpu(), F.pixel_shuffle(x, 2))
    assert torch.equal(K.pixel_unshuffle2(F.pixel_shuffle(x, 2).contiguous().cuda()).cpu(), x)
    a, b = torch.randn(3, 7, 9, generator=gen), torch.randn(3, 7, 9, generator=gen)
    al = torch.tensor([0.37])
    assert maxdiff(K.axpby_dev(a.cuda(), al.cuda(), b.cuda()), a * al + b) < 1e-6
    assert maxdiff(K.axpby_dev(a.cuda(), al.cuda()), a * al) < 1e-6
    assert abs(K.dot(a.cuda(), b.cuda()).item() - (a.double() * b.double()).sum().item()) < 1e-4


# ------------------------------------------------------------------ a15 MDTA core pieces vs torch
@pytest.mark.parametrize('C,heads,HW', [(16, 2, (8, 12)), (48, 1, (16, 16)), (192, 2, (8, 8))])
def test_mdta_core_vs_torch(K, C, heads, HW):
    """Gram -> softmax weights -> attn v, and the backward weights W with d[q;k] = W [q;k], against autograd."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(C + heads)
    N, (H, W) = 2, HW
    c = C // heads
    qkv = torch.randn(N, 3 * C, H, W, generator=gen, requires_grad=True)
    temp = (1 + 0.3 * torch.randn(heads, 1, 1, generator=gen)).requires_grad_(True)
    q, k, v = qkv.chunk(3, dim=1)
    qh = F.normalize(q.reshape(N, heads, c, H * W), dim=-1)
    kh = F.normalize(k.reshape(N, heads, c, H * W), dim=-1)
    attn = ((qh @ kh.transpose(-2, -1)) * temp).softmax(dim=-1)
    out = (attn @ v.reshape(N, heads, c, H * W)).reshape(N, C, H, W)
    go = torch.randn(N, C, H, W, generator=gen)
    out.backward(go)
    prev = K.MATH
    K.set_math('f32')
    try:
        d = qkv.detach().cuda()
        ss = K.row_sumsq(d, 2 * C)
        assert maxdiff(ss, (qkv.detach()[:, :2 * C] ** 2).sum(dim=(2, 3))) < 1e-3
        G = K.conv_wgrad(d[:, C:2 * C], d[:, :C], C, C, 1, per_image=True).view(N, C, C)
        A, AT = K.mdta_softmax(G, ss, temp.detach().cuda(), heads)
        Cp = A.shape[-1]
        for h in range(heads):
            blk = A[:, h * c:(h + 1) * c, h * c:(h + 1) * c]
            assert maxdiff(blk, attn[:, h]) < 2e-6
        assert torch.equal(A.transpose(1, 2), AT)
        o = K.conv_forward(d[:, 2 * C:], AT, Cp, C, 1, wp_ns=Cp * Cp)
        assert maxdiff(o, out) < 2e-5
        gd = go.cuda()
        dA = K.conv_wgrad(d[:, 2 * C:], gd, C, C, 1, per_image=True).view(N, C, C)
        Wm, dtemp = K.mdta_bwd(G, ss, temp.detach().cuda(), A, dA, heads)
        Wp = Wm.shape[-1]
        dqkv = torch.empty_like(d)
        K.conv_forward(gd, A, Cp, C, 1, wp_ns=Cp * Cp, out=dqkv[:, 2 * C:])
        K.conv_forward(d[:, :2 * C], Wm, Wp, 2 * C, 1, wp_ns=Wp * Wp, out=dqkv[:, :2 * C])
        assert maxdiff(dqkv, qkv.grad) < 3e-5 * max(1.0, qkv.grad.abs().max().item())
        assert maxdiff(dtemp, temp.grad) < 1e-4 * max(1.0, temp.grad.abs().max().item())
    finally:
        K.set_math(prev)


# ------------------------------------------------------------------ a16 / a17 blocks vs reference goldens
@pytest.mark.parametrize('kind', ['BiasFree', 'WithBias'])
def test_transformer_block_vs_reference_golden(R, kind):
    g = gold('restormer_per_op')
    tag = 'tblock_' + kind
    P = gparams(g, tag)
    out, saved = R.tblock_fwd(dev(g[tag + '_x']), P, 4, kind)
    assert maxdiff(out, T(g[tag + '_y'])) < 1e-4
    dx, G = R.tblock_bwd(dev(g[tag + '_go']), P, 4, kind, saved)
    assert maxdiff(dx, T(g[tag + '_gx'])) < 1e-4 * max(1.0, np.abs(g[tag + '_gx']).max())
    check_grads(g, tag, G, tol=2e-4)


def test_fusion_block_vs_reference_golden(R):
    g = gold('restormer_per_op')
    P = gparams(g, 'fblock')
    out, saved = R.fblock_fwd(dev(g['fblock_x']), P, 2, 'WithBias')
    assert maxdiff(out, T(g['fblock_y'])) < 1e-4
    dx, G = R.fblock_bwd(dev(g['fblock_go']), P, 2, 'WithBias', saved)
    assert maxdiff(dx, T(g['fblock_gx'])) < 1e-4 * max(1.0, np.abs(g['fblock_gx']).max())
    check_grads(g, 'fblock', G, tol=2e-4)


def test_down_up_sample_vs_reference_golden(R):
    g = gold('restormer_per_op')
    for tag, fwd, bwd in (('down', R.down_fwd, R.down_bwd), ('up', R.up_fwd, R.up_bwd)):
        w = dev(g[tag + '_p_body.0.weight'])
        x = dev(g[tag + '_x'])
        assert maxdiff(fwd(x, w), T(g[tag + '_y'])) < 2e-5
        dx, dw = bwd(dev(g[tag + '_go']), x, w)
        assert maxdiff(dx, T(g[tag + '_gx'])) < 5e-5
        assert maxdiff(dw, T(g[tag + '_g_body.0.weight'])) < 2e-4 * max(1.0, np.abs(g[tag + '_g_body.0.weight']).max())


# ------------------------------------------------------------------ a18 whole network
CASES = [('restormer_d8_128', dict()),
         ('restormer_d8_128_biasfree_b2', dict(LayerNorm_type='BiasFree', num_blocks=[1, 2, 1, 1])),
         ('restormer_d8_64_wrap_bias', dict(bias=True)),
         ('restormer_d16_120x100_pad', dict(dim=16, nf=16))]


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_vs_reference_golden(R, name, kw):
    g = gold(name)
    cfg = RO.default_cfg(**kw)
    seed = int(g['seed'])
    P = RO.synth_params(cfg, seed=seed)
    lq, gt, ref = NO.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=4321 + seed)
    Pc = {k: v.cuda().contiguous() for k, v in P.items()}
    out, saved = R.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
    sv_masa = saved[6]
    index, index_all, soft_att = sv_masa[4], sv_masa[7], sv_masa[8]
    assert np.array_equal(index.cpu().numpy().reshape(g['index'].shape[0], -1), g['index'].reshape(g['index'].shape[0], -1))
    assert np.array_equal(index_all.cpu().numpy().reshape(g['index_all'].shape), g['index_all'])
    assert maxdiff(soft_att.reshape(g['soft_att'].shape), T(g['soft_att'])) < 1e-5
    assert maxdiff(out, T(g['out'])) < 1e-4
    from textualdegremoval_amd import kernels as K
    loss, dpred = K.l1_loss(out.contiguous(), gt.cuda(), 1.0)
    assert abs(loss.item() - float(g['loss'])) < 2e-6
    G = R.net_bwd(dpred, Pc, cfg, saved)
    names = list(P.keys())
    assert set(G.keys()) == set(names)
    gn = np.array([G[k].double().norm().item() for k in names])
    assert np.allclose(gn, g['grad_norm'], rtol=5e-3, atol=1e-5), np.abs(gn - g['grad_norm']).max()
    for i, k in enumerate(names):
        s = G[k].detach().reshape(-1)
        step = max(1, s.numel() // 8)
        smp = s[::step][:8].cpu().numpy()
        assert np.abs(smp - g['grad_sample'][i, :len(smp)]).max() < 2e-4 * max(1.0, g['grad_norm'][i]), k


def test_module_autograd_matches_oracle(R):
    """nn.Module drop-in (define_network) forward + autograd backward == oracle autograd on the same inputs."""
    from textualdegremoval_amd.models.archs import define_network
    cfg = RO.default_cfg(LayerNorm_type='BiasFree')
    P = RO.synth_params(cfg, seed=7)
    net = define_network(dict(type='RestormerRefFusion', **cfg)).cuda()
    net.load_state_dict(P, strict=True)
    lq, gt, ref = NO.synth_pair(1, 128, 128, seed=77)
    out = net(lq.cuda(), ref.cuda())
    (out - gt.cuda()).abs().mean().backward()
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oo = RO.restormer_ref_forward(Po, cfg, lq, ref)
    (oo - gt).abs().mean().backward()
    assert maxdiff(out, oo) < 1e-4
    for k, p in net.named_parameters():
        ref_g = Po[k].grad
        assert maxdiff(p.grad, ref_g) < 3e-4 * max(1e-3, ref_g.abs().max().item()) + 1e-7, k


def test_train_step_matches_oracle_trainer():
    """RefGuidedImageCleanModel.optimize_parameters with a Restormer-ref net_g: eager, eager, graph capture, replay."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd.models import create_model
    cfg = RO.default_cfg()
    opt = {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 1, 'dist': False, 'is_train': True,
        'network_g': dict(type='RestormerRefFusion', **cfg), 'path': {},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                  'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'val': {}, 'scale': 1,
    }
    model = create_model(opt)
    P = RO.synth_params(cfg, seed=5)
    model.net_g.load_state_dict(P, strict=True)
    tr = NO.OracleTrainer(P, cfg, forward_fn=RO.restormer_ref_forward)
    lq, gt, ref = NO.synth_pair(1, 128, 128, seed=55)
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
    assert maxdiff(model.output, out_o) < 1e-4

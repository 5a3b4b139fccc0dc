"""GPU parity of blocks / encoder / whole NAFNet-ref forward+backward against
(a) golden vectors produced by the reference itself and (b) the oracle on the
same seeded inputs.  Path target: 1e-4 max-abs on fp32 outputs (north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', params=['bx3', 'f32', 'hx2'])
def E(request):
    """whole-network parity under both matrix-core arithmetic modes (kernels.MATH)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import engine, kernels
    prev = kernels.MATH
    kernels.set_math(request.param)
    yield engine
    kernels.set_math(prev)


def gold(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def cuda_params(P):
    return {k: v.detach().cuda().contiguous() for k, v in P.items()}


def test_nafblock_vs_reference_golden(E):
    g = gold('per_op')
    P = {str(k): T(g['naf_p_' + str(k)]) for k in g['naf_names']}
    out, saved = E.naf_fwd(T(g['naf_x']).cuda(), cuda_params(P))
    assert maxdiff(out, T(g['naf_y'])) < 5e-5
    dx, G = E.naf_bwd(T(g['naf_go']).cuda(), cuda_params(P), saved)
    assert maxdiff(dx, T(g['naf_gx'])) < 5e-5
    for k in P:
        ref = T(g['naf_g_' + k])
        assert maxdiff(G[k].view_as(ref), ref) < 1e-4 * max(1.0, ref.abs().max().item()), k


def test_nafblock_sliced_output_matches_oracle(E):
    """last fusion block: only the first c/2 output channels are kept (:719)."""
    c, N, H, W = 32, 2, 16, 24
    cfgP = {k[len('b.'):]: v for k, v in O.synth_params(O.default_cfg(), seed=9).items() if False}
    g = torch.Generator().manual_seed(5)
    shapes = dict(beta=(1, c, 1, 1), gamma=(1, c, 1, 1))
    P = {}
    for name, shp in [('beta', (1, c, 1, 1)), ('gamma', (1, c, 1, 1)), ('conv1.weight', (2 * c, c, 1, 1)), ('conv1.bias', (2 * c,)),
                      ('conv2.weight', (2 * c, 1, 3, 3)), ('conv2.bias', (2 * c,)), ('conv3.weight', (c, c, 1, 1)),
                      ('conv3.bias', (c,)), ('sca.1.weight', (c, c, 1, 1)), ('sca.1.bias', (c,)),
                      ('conv4.weight', (2 * c, c, 1, 1)), ('conv4.bias', (2 * c,)), ('conv5.weight', (c, c, 1, 1)),
                      ('conv5.bias', (c,)), ('norm1.weight', (c,)), ('norm1.bias', (c,)), ('norm2.weight', (c,)),
                      ('norm2.bias', (c,))]:
        P[name] = (torch.randn(shp, generator=g) * 0.2 + (1.0 if name.endswith('norm1.weight') or name.endswith('norm2.weight') else 0.0))
    x = torch.randn(N, c, H, W, generator=g)
    go = torch.randn(N, c // 2, H, W, generator=g)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.naf_block(xr, Pr, '')[:, :c // 2]
    ref.backward(go)
    out, saved = E.naf_fwd(x.cuda(), cuda_params(P), c_out=c // 2)
    assert maxdiff(out, ref) < 5e-5
    dx, G = E.naf_bwd(go.cuda(), cuda_params(P), saved)
    assert maxdiff(dx, xr.grad) < 5e-5
    for k in P:
        r = Pr[k].grad
        assert maxdiff(G[k].view_as(r), r) < 1e-4 * max(1.0, r.abs().max().item()), k


def test_masa_encoder_vs_reference_golden(E):
    g = gold('per_op')
    P = {str(k): T(g['enc_p_' + str(k)]) for k in g['enc_names']}
    Pc = cuda_params(P)
    feats, saved = E.encoder_fwd(T(g['enc_x']).cuda(), Pc, '', [1, 1, 1, 1])
    for i, f in enumerate(feats):            # activations here are O(100): relative tolerance
        ref = T(g[f'enc_f{i}'])
        assert maxdiff(f, ref) < 3e-6 * max(1.0, ref.abs().max().item())
    # loss = sum_i (i+1) * mean(f_i^2)  ->  df_i = 2 (i+1) f_i / numel
    dfe = [(2.0 * (i + 1) / f.numel()) * f for i, f in enumerate(feats)]
    G = {}
    E.encoder_bwd(dfe, Pc, '', [1, 1, 1, 1], saved, G)
    for k in P:
        ref = T(g['enc_g_' + k])
        assert maxdiff(G[k].view_as(ref), ref) < 3e-5 * max(1.0, ref.abs().max().item()), k


CASES = [('net_w8_256_b2_clear', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_128_wrap', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_256_b2', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_120x100_pad', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_cfg1_w16_128', dict(width=16, nf=16, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])),
         # the shipped NAFNet YAML's widths (width = nf = 64: up to 2048 channels in the middle fusion block), make_golden.py yaml64
         ('net_yaml_w64_128', dict(width=64, nf=64, enc_blk_nums=[1, 1, 1, 3], dec_blk_nums=[1, 1, 1, 1], middle_blk_num=1,
                                   ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 1])),
         # ref of another size than lq: two separate masa_enc passes, block diameter from the ref size (validation /
         # inference path of the reference, image_restoration_ref_model.py:286-330)
         ('net_w8_256_ref384', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_128_ref256_wrap', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_200x136_ref300', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]))]
REF_HW = {'net_w8_256_ref384': (384, 384), 'net_w8_128_ref256_wrap': (256, 256), 'net_w8_200x136_ref300': (300, 300)}


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_vs_reference_golden(E, name, kw):
    from textualdegremoval_amd import kernels as K
    g = gold(name)
    cfg = O.default_cfg(**kw)
    seed = int(g['seed'])
    P = O.synth_params(cfg, seed=seed)
    Pc = cuda_params(P)
    lq, gt, ref = O.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=1234 + seed, ref_hw=REF_HW.get(name))
    out, saved = E.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
    sv_masa = saved[6]
    index, index_all, soft_att = sv_masa[4], sv_masa[7], sv_masa[8]
    gi = g['index'][..., 0] if g['index'].ndim == 3 else g['index']
    assert np.array_equal(index.cpu().numpy().reshape(gi.shape), gi)
    # hard-attention protocol (SURVEY hard part 2): the arg-max may only differ from the reference
    # where the reference's own top-1/top-2 gap is below 1e-5 (summation-order noise); such cases are
    # then compared teacher-forced (reference indices fed to the transfer kernels).
    ia = index_all.cpu().numpy().reshape(g['index_all'].shape)
    mism = ia != g['index_all']
    # (the zero-padded ref300 case has whole blocks of EXACT ties -- constant features over the padding -- where any
    # index is "the" arg-max; every flip must still sit on a tie of the reference's own scores)
    assert mism.mean() <= (0.10 if name == 'net_w8_200x136_ref300' else 0.005), f'fine-search index agreement {1 - mism.mean()}'
    assert (g['fine_gap'].reshape(mism.shape)[mism] < 1e-5).all(), 'index flip at a non-tie'
    if mism.any():
        forced = torch.from_numpy(g['index_all'].reshape(index_all.shape)).int().cuda()
        orig = K.fine_argmax

        def teacher(dots, invq, invk, B, Pn, R):
            _, _ = orig(dots, invq, invk, B, Pn, R)
            val = (dots.view(B, Pn, R) * invq.view(B, Pn, 1) * invk.view(B, 1, R)).gather(2, forced.long().view(B, Pn, 1))
            return forced, val.view(B, Pn).contiguous()
        K.fine_argmax = teacher
        try:
            out, saved = E.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
        finally:
            K.fine_argmax = orig
        soft_att = saved[6][8]
    assert maxdiff(soft_att.view(-1), T(g['soft_att']).reshape(-1)) < 1e-5
    assert maxdiff(out, T(g['out'])) < 1e-4
    loss, dpred = K.l1_loss(out.contiguous(), gt.cuda().contiguous())
    assert abs(loss.item() - float(g['loss'])) < 2e-6
    G = E.net_bwd(dpred, Pc, cfg, saved)
    assert set(G.keys()) == set(P.keys())
    gn = np.array([G[k].double().norm().item() for k in P])
    assert np.allclose(gn, g['grad_norm'], rtol=5e-3, atol=5e-6), np.abs(gn - g['grad_norm']).max()
    tot = np.sqrt((gn ** 2).sum())
    assert abs(tot - float(g['total_grad_norm'])) < 2e-4 * float(g['total_grad_norm'])
    # element samples of every gradient (same strided sampling as make_golden.sample)
    for i, k in enumerate(P):
        f = G[k].reshape(-1)
        step = max(1, f.numel() // 8)
        s = f[::step][:8].cpu().numpy()
        ref_s = g['grad_sample'][i][:len(s)]
        assert np.allclose(s, ref_s, rtol=2e-3, atol=2e-6 + 2e-4 * np.abs(g['grad_sample'][i]).max()), k


def test_module_surface_forward_backward_and_state_dict(E):
    """nn.Module surface: load a reference-keyed state dict, forward(inp, ref), autograd backward."""
    from textualdegremoval_amd.models.archs import define_network
    g = gold('net_w8_128_wrap')
    kw = dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    cfg = O.default_cfg(**kw)
    net = define_network(dict(type='NAFNetRefFusion', enc_blk_nums=[1, 1, 1, 1], dec_blk_nums=[1, 1, 1, 1], **kw))
    net.load_state_dict(O.synth_params(cfg, seed=int(g['seed'])), strict=True)
    net = net.cuda()
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1234 + int(g['seed']))
    out = net(lq.cuda(), ref.cuda())
    assert maxdiff(out, T(g['out'])) < 1e-4
    (out - gt.cuda()).abs().mean().backward()
    gn = np.array([p.grad.double().norm().item() for p in net.parameters()])
    assert np.allclose(gn, g['grad_norm'], rtol=5e-3, atol=5e-6)


def test_side_stream_weight_gradients_match(E):
    """kernels.SIDE_WGRAD: weight gradients on a parallel stream give the same results (bit-identical except where the
    MASA transfer backward accumulates with float atomics, whose order varies from run to run anyway)."""
    from textualdegremoval_amd import kernels as K
    cfg = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    P = cuda_params(O.synth_params(cfg, seed=2))
    lq, gt, ref = O.synth_pair(2, 128, 128, seed=12)
    res = []
    for side in (False, True):
        prev, K.SIDE_WGRAD = K.SIDE_WGRAD, side
        grp, E.GROUP_LEAVES = E.GROUP_LEAVES, False      # (same kernels on the same operands: the grouped launch of the deferred leaves
        try:                                              #  cuts the pixel sums differently, tests/test_hip_wgrad1x1_group.py)
            out, saved = E.net_fwd(P, cfg, lq.cuda(), ref.cuda())
            loss, dpred = K.l1_loss(out.contiguous(), gt.cuda(), 1.0)
            G = E.net_bwd(dpred, P, cfg, saved)
            torch.cuda.synchronize()
            res.append({k: v.clone() for k, v in G.items()})
        finally:
            K.SIDE_WGRAD = prev
            E.GROUP_LEAVES = grp
    assert set(res[0]) == set(res[1])
    for k in res[0]:
        if k.startswith('masa_enc.'):
            assert maxdiff(res[0][k], res[1][k]) <= 1e-5 * max(1e-6, res[0][k].abs().max().item()), k
        else:
            assert torch.equal(res[0][k], res[1][k]), k


def test_deferred_leaf_weight_gradients_match(E):
    """engine.DEFER_WGRAD (default without a gradient exchange): the conv1 / conv4 / conv5 weight gradients of every NAFBlock are queued
    and run on a second stream beside the MASA-encoder backward -- same kernels on the same operands, so every tensor outside the
    MASA encoder (atomics in the transfer backward) is bit-identical to the single-stream backward; the set of keys is the same and
    the deferred operands were still alive (a recycled buffer would show up as a wrong gradient)."""
    from textualdegremoval_amd import kernels as K
    cfg = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    P = cuda_params(O.synth_params(cfg, seed=2))
    lq, gt, ref = O.synth_pair(2, 128, 128, seed=12)
    res = []
    for defer in (False, True, True):
        prev, E.DEFER_WGRAD = E.DEFER_WGRAD, defer
        grp, E.GROUP_LEAVES = E.GROUP_LEAVES, False      # (per-leaf launches: the grouped form is compared in tests/test_hip_wgrad1x1_group.py)
        try:
            out, saved = E.net_fwd(P, cfg, lq.cuda(), ref.cuda())
            loss, dpred = K.l1_loss(out.contiguous(), gt.cuda(), 1.0)
            junk = [torch.full((1 << 20,), float('nan'), device='cuda') for _ in range(4)]      # churn the allocator around the backward
            G = E.net_bwd(dpred, P, cfg, saved)
            del junk
            torch.cuda.synchronize()
            res.append({k: v.clone() for k, v in G.items()})
        finally:
            E.DEFER_WGRAD = prev
            E.GROUP_LEAVES = grp
    assert E._late is None
    for other in res[1:]:
        assert set(res[0]) == set(other)
        for k in res[0]:
            if k.startswith('masa_enc.'):
                assert maxdiff(res[0][k], other[k]) <= 1e-5 * max(1e-6, res[0][k].abs().max().item()), k
            else:
                assert torch.equal(res[0][k], other[k]), k


@pytest.mark.parametrize('name', ['net_w8_256_b2_clear', 'net_w8_128_wrap', 'net_w8_256_ref384'])
def test_deterministic_mode_is_bit_reproducible_and_matches_golden(E, name):
    """TDR_DETERMINISTIC=1 (kernels.DETERMINISTIC): the MASA transfer backward accumulates in 64-bit fixed point -- the only
    order-dependent reduction of the step -- so repeated backward passes give bit-identical gradients; the values still
    match the reference's golden gradient norms, and the float-atomic default to its usual ~1e-6."""
    from textualdegremoval_amd import kernels as K
    g = gold(name)
    kw = dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    cfg = O.default_cfg(**kw)
    seed = int(g['seed'])
    P = O.synth_params(cfg, seed=seed)
    Pc = cuda_params(P)
    lq, gt, ref = O.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=1234 + seed, ref_hw=REF_HW.get(name))
    runs = []
    for det in (True, True, True, False):
        prev, K.DETERMINISTIC = K.DETERMINISTIC, det
        try:
            out, saved = E.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
            loss, dpred = K.l1_loss(out.contiguous(), gt.cuda().contiguous())
            G = E.net_bwd(dpred, Pc, cfg, saved)
            torch.cuda.synchronize()
            runs.append({k: v.clone() for k, v in G.items()})
        finally:
            K.DETERMINISTIC = prev
    for k in P:
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
        ref_g = runs[3][k]
        assert maxdiff(runs[0][k], ref_g) <= 2e-5 * max(ref_g.abs().max().item(), 1e-7), k
    gn = np.array([runs[0][k].double().norm().item() for k in P])
    assert np.allclose(gn, g['grad_norm'], rtol=5e-3, atol=5e-6)

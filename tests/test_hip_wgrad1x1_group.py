"""Grouped 1x1 weight gradients (kernels.wgrad1x1_group -> tdr_wgrad1x1_group, csrc/tdr_wgrad_1x1.hip): the leaf weight / bias gradients
of all NAFBlocks of one level in ONE launch + ONE fixed-order reduction.  Checked against float64 per problem (plain, gated, ragged
channel tiles, image splits), against the per-problem launches, bit-for-bit across repeated calls (deterministic), and -- through
engine.net_bwd -- with grouping on against grouping off.  Replaces autograd's weight gradients of conv1 / conv4 / conv5 of the
reference's NAFBlocks (models/archs/network_nafnet_guided_arch.py:183-205,216-238)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels as K
    prev = K.MATH
    K.set_math('bx3')
    yield K
    K.set_math(prev)


@pytest.mark.parametrize('nprob,N,Cin,Cout,H,W,gate', [
    (5, 4, 256, 512, 16, 16, False),      # many problems: one image per workgroup
    (3, 2, 128, 256, 32, 32, True),       # gated operand (conv5)
    (2, 1, 96, 72, 40, 40, False),        # ragged channel tiles, one image: the images are split
    (1, 3, 160, 136, 8, 20, False),       # a single problem
    (4, 2, 128, 128, 24, 12, True),
    (6, 2, 64, 128, 32, 32, False),       # 64 input channels: 128 x 64 output tiles
    (3, 4, 64, 192, 16, 24, True),
    (2, 2, 288, 320, 16, 16, False),      # 256 x 256 output tiles, ragged in both directions
    (2, 2, 256, 256, 16, 8, True)])
def test_group_vs_fp64_and_single_launches(K, nprob, N, Cin, Cout, H, W, gate):
    torch.manual_seed(nprob * 7 + Cin)
    reqs = []
    for _ in range(nprob):
        x = torch.randn(N, Cin * (2 if gate else 1), H, W, device='cuda') * 3e-3
        d = torch.randn(N, Cout, H, W, device='cuda') * 2e-5
        reqs.append((x, d, Cout, Cin, gate))
    assert K.wgrad1x1_group_key(*reqs[0]) is not None
    assert len({K.wgrad1x1_group_key(*r) for r in reqs}) == 1
    out = K.wgrad1x1_group(reqs, seq=900 + nprob)
    again = K.wgrad1x1_group(reqs, seq=900 + nprob)
    for (x, d, *_), (g, db), (g2, db2) in zip(reqs, out, again):
        xe = (x[:, :Cin] * x[:, Cin:]) if gate else x
        ref = torch.einsum('nkp,ncp->kc', d.double().flatten(2), xe.double().flatten(2))
        rb = d.double().sum((0, 2, 3))
        assert (g.double().view(Cout, Cin) - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
        assert (db.double() - rb).abs().max().item() < 2e-6 * rb.abs().max().item()
        assert torch.equal(g, g2) and torch.equal(db, db2)                       # fixed-order reduction: deterministic
        gs, dbs = K.conv_wgrad(x, d, Cout, Cin, 1, gate=gate, want_db=True)     # the per-problem launch (another split of the pixels)
        assert (g - gs).abs().max().item() < 2e-6 * ref.abs().max().item()
        assert (db - dbs).abs().max().item() < 2e-6 * rb.abs().max().item()


def test_network_backward_grouped_equals_ungrouped(K):
    """the whole guided network: parameter gradients with the deferred leaves grouped against the per-leaf launches"""
    from oracle import nafnet_ref_oracle as O
    from textualdegremoval_amd import engine as E
    cfg = O.default_cfg(width=32, nf=32, enc_blk_nums=[1, 1, 1, 2], ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    P = {k: v.cuda() for k, v in O.synth_params(cfg, seed=5).items()}
    lq, gt, ref = (t.cuda() for t in O.synth_pair(2, 128, 128, seed=6))
    res = {}
    was = E.GROUP_LEAVES
    try:
        for mode in (True, False):
            E.GROUP_LEAVES = mode
            out, saved = E.net_fwd(P, cfg, lq, ref)
            _, dpred = K.l1_loss(out.contiguous(), gt)
            res[mode] = {k: v.clone() for k, v in E.net_bwd(dpred, P, cfg, saved).items()}
    finally:
        E.GROUP_LEAVES = was
    assert res[True].keys() == res[False].keys()
    worst = 0.0
    for k, g in res[False].items():
        worst = max(worst, (res[True][k] - g).abs().max().item() / max(g.abs().max().item(), 1e-30))
    # the grouped launch cuts the pixel sum of a weight gradient differently (whole images instead of 32 slices): rounding only
    assert worst < 5e-6, worst


def test_finishing_reductions_batched_are_bit_identical(K):
    """the table-driven forms of the three small finishing reductions (kernels.pair_sum_partials_multi / dw_param_finish_multi /
    scaled_conv_param_grads_multi) write, per problem, exactly what the single-problem entry points write -- problems of different
    shapes in ONE launch"""
    lib = __import__('textualdegremoval_amd._lib', fromlist=['load']).load()
    g = torch.Generator().manual_seed(11)
    # LayerNorm-gradient partials [nparts][2][C]
    items = [(torch.randn(nparts * 2 * Cc, generator=g).cuda(), nparts, Cc)
             for nparts, Cc, n in ((256, 256, 5), (37, 96, 3), (1024, 32, 2), (1, 128, 1)) for _ in range(n)]
    for (w, nparts, Cc), (gw, gb) in zip(items, K.pair_sum_partials_multi(items, seq=900)):
        rw, rb = torch.empty(Cc, device='cuda'), torch.empty(Cc, device='cuda')
        K.check(lib.tdr_pair_sum_partials(w.data_ptr(), nparts, Cc, rw.data_ptr(), rb.data_ptr(), 0, K._stream()), 'pair_sum')
        assert torch.equal(gw, rw) and torch.equal(gb, rb)
        assert (gw.double().cpu() - w.view(nparts, 2, Cc)[:, 0].double().sum(0).cpu()).abs().max() < 1e-3
    # depthwise parameter partials
    items = [(torch.randn(int(lib.tdr_dwsg_ws_floats(N, Cc, H, W)), generator=g).cuda(), N, Cc, H, W)
             for N, Cc, H, W, n in ((4, 64, 64, 64, 4), (2, 24, 32, 48, 3), (4, 256, 16, 16, 2)) for _ in range(n)]
    for (w, N, Cc, H, W), (dw, db) in zip(items, K.dw_param_finish_multi(items, seq=901)):
        rw, rb = torch.empty(2 * Cc, 1, 3, 3, device='cuda'), torch.empty(2 * Cc, device='cuda')
        K.check(lib.tdr_dw_param_finish(w.data_ptr(), N, Cc, H, W, rw.data_ptr(), rb.data_ptr(), K._stream()), 'dw_finish')
        assert torch.equal(dw, rw) and torch.equal(db, rb)
    # conv5 / gamma parameter gradients
    items = [tuple(torch.randn(*shp, generator=g).cuda() for shp in ((Cout, Cin), (Cout,), (Cout, Cin), (Cout,), (Cout,)))
             for Cout, Cin, n in ((256, 256, 6), (128, 256, 2), (40, 72, 3)) for _ in range(n)]
    for it, r3 in zip(items, K.scaled_conv_param_grads_multi(items, seq=902)):
        ref = K.scaled_conv_param_grads(*it)
        assert all(torch.equal(a, b) for a, b in zip(r3, ref))


def test_network_backward_with_batched_finishers_is_bit_identical(K):
    """engine.BATCH_FINISH on / off through the whole guided network (deferred leaves): the same gradients bit for bit"""
    from oracle import nafnet_ref_oracle as O
    from textualdegremoval_amd import engine as E
    cfg = O.default_cfg(width=32, nf=32, enc_blk_nums=[1, 1, 1, 3], ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    P = {k: v.cuda() for k, v in O.synth_params(cfg, seed=7).items()}
    lq, gt, ref = (t.cuda() for t in O.synth_pair(2, 128, 128, seed=8))
    res = {}
    was = E.BATCH_FINISH, E.FORCE_DP_SCHEDULE, K.DETERMINISTIC
    K.DETERMINISTIC = True                 # (the MASA transfer's scatter in fixed point: otherwise two runs of ONE setting differ in the last bit)
    try:
        for dp in (False, True):
            for mode in (True, False):
                E.BATCH_FINISH, E.FORCE_DP_SCHEDULE = mode, dp
                out, saved = E.net_fwd(P, cfg, lq, ref)
                _, dpred = K.l1_loss(out.contiguous(), gt)
                res[dp, mode] = {k: v.clone() for k, v in E.net_bwd(dpred, P, cfg, saved).items()}
            assert res[dp, True].keys() == res[dp, False].keys()
            for k, gr in res[dp, False].items():
                assert torch.equal(res[dp, True][k], gr), (dp, k)
    finally:
        E.BATCH_FINISH, E.FORCE_DP_SCHEDULE, K.DETERMINISTIC = was

"""The device-resident step guard (include/tdr.h TdrStepGuard, textualdegremoval_amd/optim.py): a non-finite gradient
norm skips the optimiser step and halves the loss scale of the fp16-split backward pass; AdamW's step count / bias
corrections only advance on applied steps; frozen parameter groups (the reference's fix_iterations branch,
image_restoration_ref_model.py:205-212), the `optim_g.type: Adam` branch (:176-178) and the EMA update
(base_model.py:54-62) on the multi-tensor kernels -- each against torch on the CPU."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


SHAPES = [(5000,), (7, 3, 3, 3), (1, 16, 1, 1), (33,), (4097,)]


def _pair(coupled=False, frozen=()):
    from textualdegremoval_amd.optim import FusedClipAdamW
    ps = [torch.nn.Parameter(rnd(*s, seed=i)) for i, s in enumerate(SHAPES)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    groups = [{'params': ref[:2], 'lr': 2e-4}, {'params': ref[2:], 'lr': 1e-4}]
    cls = torch.optim.Adam if coupled else torch.optim.AdamW
    opt_ref = cls(groups, lr=2e-4, weight_decay=1e-2, betas=(0.9, 0.999))
    gp = [torch.nn.Parameter(p.detach().cuda()) for p in ps]
    opt = FusedClipAdamW([{'params': gp[:2], 'lr': 2e-4}, {'params': gp[2:], 'lr': 1e-4}], lr=2e-4, weight_decay=1e-2,
                         betas=(0.9, 0.999), max_norm=0.01, use_grad_clip=True, coupled_decay=coupled)
    opt.set_frozen_groups(set(frozen))
    return ref, opt_ref, gp, opt


def _set_grads(ref, gp, step, frozen_from=None, poison=None):
    for i, (a, b) in enumerate(zip(ref, gp)):
        gr = rnd(*a.shape, seed=100 + 10 * step + i, scale=0.01)
        dev = gr.clone()
        if poison is not None and i == 1:
            dev.view(-1)[5] = poison
        a.grad = None if (frozen_from is not None and i >= frozen_from) else gr.clone()
        b.grad = dev.cuda() if b.grad is None else b.grad.copy_(dev.cuda())


def test_non_finite_norm_skips_the_step_and_halves_the_scale():
    ref, opt_ref, gp, opt = _pair()
    _set_grads(ref, gp, 0)
    opt.prepare()                                         # builds the tables and the guard
    opt.guard.write(scale=1024.0, max_scale=1024.0)
    before = [p.detach().clone() for p in gp]
    for poison in (float('inf'), float('nan')):
        _set_grads(ref, gp, 0, poison=poison)
        opt.step()
        for a, b in zip(before, gp):
            assert torch.equal(a, b.detach())             # nothing moved, moments included
    g = opt.guard.read()
    assert (g.skipped, g.step, g.finite) == (2, 0, 0) and g.scale == 256.0 and g.inv_scale == 1.0 / 256.0
    assert all(float(opt.state[p]['exp_avg'].abs().max()) == 0.0 for p in gp)
    # the next finite steps are AdamW steps 1, 2, 3 exactly (bias corrections from the applied-step count)
    for step in range(3):
        _set_grads(ref, gp, step)
        torch.nn.utils.clip_grad_norm_(ref, 0.01)
        opt_ref.step(); opt.step()
    for a, b in zip(ref, gp):
        assert (b.detach().cpu() - a.detach()).abs().max().item() < 1e-6
    g = opt.guard.read()
    assert (g.skipped, g.step, g.finite) == (2, 3, 1)
    assert int(opt.state_dict()['state'][0]['step']) == 3


def test_without_a_loss_scale_there_is_no_verdict():
    """TDR_MATH=bx3 / f32 (the reference's arithmetic): the struct only counts steps -- a non-finite norm does NOT skip the step
    (torch's clip_grad_norm_ + AdamW.step() apply it too, image_restoration_ref_model.py:276-279) and the scale never moves"""
    ref, opt_ref, gp, opt = _pair()
    _set_grads(ref, gp, 0)
    opt.prepare()
    opt.guard.set_never_skip(True)
    assert opt.guard.never_skip
    _set_grads(ref, gp, 0, poison=float('inf'))
    opt.step()
    g = opt.guard.read()
    assert (g.skipped, g.step, g.finite) == (0, 1, 1) and g.scale == 1.0
    assert any(not torch.isfinite(p.detach()).all() for p in gp)         # the poisoned update went through
    opt.guard.set_never_skip(False)
    assert opt.guard.read().growth_interval == 1000


def test_scale_grows_back_after_the_growth_interval():
    ref, opt_ref, gp, opt = _pair()
    _set_grads(ref, gp, 0)
    opt.prepare()
    opt.guard.growth_interval = 2
    opt.guard.write(scale=64.0, max_scale=256.0)
    seen = []
    for step in range(6):
        _set_grads(ref, gp, step)
        opt.step()
        seen.append(opt.guard.read().scale)
    assert seen == [64.0, 128.0, 128.0, 256.0, 256.0, 256.0]


def test_frozen_group_matches_torch_with_grad_none():
    ref, opt_ref, gp, opt = _pair(frozen={1})
    for step in range(3):
        _set_grads(ref, gp, step, frozen_from=2)
        torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], 0.01)
        opt_ref.step(); opt.step()
    for a, b in zip(ref, gp):
        assert (b.detach().cpu() - a.detach()).abs().max().item() < 1e-6
    for i in (2, 3, 4):                                   # frozen tensors: bit-identical to their initial values
        assert torch.equal(gp[i].detach().cpu(), rnd(*SHAPES[i], seed=i))


def test_adam_coupled_decay_matches_torch_adam():
    ref, opt_ref, gp, opt = _pair(coupled=True)
    for step in range(3):
        _set_grads(ref, gp, step)
        torch.nn.utils.clip_grad_norm_(ref, 0.01)
        opt_ref.step(); opt.step()
    for a, b in zip(ref, gp):
        assert (b.detach().cpu() - a.detach()).abs().max().item() < 1e-6


def test_model_ema_kernel_matches_torch():
    from test_hip_step import make_opt
    from textualdegremoval_amd.models import create_model
    opt = make_opt()
    opt['train']['ema_decay'] = 0.9
    model = create_model(opt)
    with torch.no_grad():
        for i, p in enumerate(model.net_g.parameters()):
            p.copy_(rnd(*p.shape, seed=i).cuda())
    want = {k: 0.9 * p.detach().cpu() for k, p in model.net_g_ema.named_parameters()}
    for k, p in model.net_g.named_parameters():
        want[k] = want[k] + (1 - 0.9) * p.detach().cpu()
    model.model_ema(0.9)
    for k, p in model.net_g_ema.named_parameters():
        assert (p.detach().cpu() - want[k]).abs().max().item() < 1e-6


@pytest.mark.parametrize('graph', ['1', '0'])
def test_overflowing_backward_is_skipped_then_recovers(monkeypatch, graph, hx2_mode):
    """A loss scale 2^22 above the surveyed one pushes gradient operands of the fp16-split backward pass out of the
    fp16 range: those steps must leave the weights untouched and halve the scale until the backward pass fits again,
    in the captured-graph step as well as the eager one; the loss stays finite throughout."""
    from test_hip_step import make_opt
    from oracle import nafnet_ref_oracle as O
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models import create_model
    assert K.MATH == 'hx2'          # (the loss-scaled backward pass exists only there)
    monkeypatch.setenv('TDR_GRAPH', graph)
    model = create_model(make_opt())
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=7)
    data = {'lq': lq, 'gt': gt, 'ref': ref}
    model.feed_train_data(data)
    model.optimize_parameters(1)
    g0 = model.optimizer_g.guard.read()
    assert g0.step == 1 and g0.skipped == 0 and g0.scale == g0.max_scale
    model.optimizer_g.guard.write(scale=g0.scale * 2.0 ** 22)       # max_scale stays: the scale only comes down
    w = copy.deepcopy({k: v.detach().clone() for k, v in model.net_g.state_dict().items()})
    skipped_seen, it = 0, 1
    while True:
        it += 1
        model.feed_train_data(data)
        model.optimize_parameters(it)
        assert model.get_current_log()['l_pix'] < 1.0
        g = model.optimizer_g.guard.read()
        if g.finite:
            break
        skipped_seen += 1
        assert all(torch.equal(w[k], v) for k, v in model.net_g.state_dict().items()), 'a skipped step moved the weights'
        assert it < 40
    assert skipped_seen >= 1 and g.skipped == skipped_seen and g.step == 2
    assert g.scale == g0.scale * 2.0 ** (22 - skipped_seen)
    assert any(not torch.equal(w[k], v) for k, v in model.net_g.state_dict().items())
    assert model.skipped_steps == skipped_seen


def test_hx2_full_range_fallback_keeps_the_verdict(monkeypatch, hx2_mode):
    """TDR_MATH=hx2 with the surveyed full-range backward (`_bwd_full_range`: no loss scale, gradients on the bf16 split): the FORWARD
    pass still runs inside the fp16 window, so the guard must keep its verdict -- a forward overflow between two surveys gives a
    non-finite norm, and that step has to be skipped instead of poisoning the weights."""
    from test_hip_step import make_opt
    from oracle import nafnet_ref_oracle as O
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models import create_model
    assert K.MATH == 'hx2'
    monkeypatch.setenv('TDR_GRAPH', '0')
    monkeypatch.setenv('TDR_RANGE_CHECK_EVERY', '0')                # between two surveys (0 = no survey steps)
    model = create_model(make_opt())
    model._bwd_full_range = True
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=7)
    data = {'lq': lq, 'gt': gt, 'ref': ref}
    model.feed_train_data(data)
    model.optimize_parameters(1)
    g = model.optimizer_g.guard
    assert not g.never_skip and g.read().scale == 1.0               # no loss scale, verdict still armed
    w = copy.deepcopy({k: v.detach().clone() for k, v in model.net_g.state_dict().items()})
    big = {k: v * 1e6 for k, v in data.items()}                     # operands beyond 65504 in the first convolutions
    model.feed_train_data(big)
    try:
        model.optimize_parameters(2)
    except RuntimeError:
        pass                                                        # (a loud non-finite loss is fine too: the weights must not move)
    gd = g.read()
    if not gd.finite:
        assert gd.skipped == 1
    assert all(torch.equal(w[k], v) and torch.isfinite(v).all() for k, v in model.net_g.state_dict().items())

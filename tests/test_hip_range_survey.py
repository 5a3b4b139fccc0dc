"""The fp16-window survey of the default arithmetic (TDR_MATH=hx2, models/image_restoration_ref_model.py::_surveyed_step):
(i) forward activations beyond the fp16 range and (ii) gradient operands 2^-40 below the window are injected; the step
must either be handled (loss scale moved / the pass taken off the fp16 split, weights protected by the step guard) or
fail loudly -- never continue silently on garbage."""
import copy

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as O

pytestmark = pytest.mark.gpu


def _model(monkeypatch, every='1'):
    from test_hip_step import make_opt
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models import create_model
    assert K.MATH == 'hx2'          # (the survey guards the fp16-split arithmetic: `hx2_mode`)
    monkeypatch.setenv('TDR_RANGE_CHECK_EVERY', every)
    model = create_model(make_opt())
    cfg = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    model.net_g.load_state_dict(O.synth_params(cfg, seed=3), strict=True)
    return model, K


def _step(model, it, data):
    model.update_learning_rate(it, warmup_iter=-1)
    model.feed_train_data(data)
    model.optimize_parameters(it)


def test_survey_reports_the_window_and_leaves_a_healthy_step_alone(monkeypatch, hx2_mode):
    model, K = _model(monkeypatch)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1234 + 3)
    _step(model, 1, {'lq': lq, 'gt': gt, 'ref': ref})
    r = model.last_range_survey
    lo, hi = model.GRAD_WINDOW
    assert r['fwd'][2] == 0 and r['grad'][2] == 0
    assert r['fwd'][1] <= model.FWD_MAX_EXP
    assert lo <= r['grad'][0] and r['grad'][1] <= hi, r
    assert K.MATH == 'hx2' and not getattr(model, '_bwd_full_range', False) and getattr(model, '_scale_shift', 0) == 0
    g = model.optimizer_g.guard.read()
    assert g.step == 1 and g.skipped == 0


def test_tiny_gradients_move_the_loss_scale_back_into_the_window(monkeypatch, hx2_mode):
    """gradient operands 2^-40 below where the surveyed scale puts them: the first survey raises the loss scale by the
    measured deficit, the second one finds the window restored; the trajectory stays on the reference's"""
    model, K = _model(monkeypatch)
    g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'trajectory.npz'))
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1234 + 3)
    data = {'lq': lq, 'gt': gt, 'ref': ref}
    model._scale_shift = -40
    monkeypatch.setenv('TDR_RANGE_CHECK_EVERY', '1000')             # corrections re-arm the survey themselves
    lo, hi = model.GRAD_WINDOW
    spans = []
    for it in range(1, 7):
        _step(model, it, data)
        if getattr(model, '_last_survey_iter', None) == it:          # this step was a survey and found nothing to correct
            break
        spans.append(model.last_range_survey['grad'])
    r = model.last_range_survey
    assert spans and spans[0][0] < lo - 15, spans                     # the injection was seen ...
    assert it <= 5 and lo <= r['grad'][0] and r['grad'][1] <= hi, (spans, r)      # ... and corrected within a few surveys
    assert model._scale_shift > -40 + 15
    assert model.optimizer_g.guard.read().skipped == 0
    # once the window holds the steps are accurate again: compare one step against a fresh model taking the same step
    ref_model, _ = _model(monkeypatch)
    ref_model.net_g.load_state_dict(model.net_g.state_dict())
    ref_model.optimizer_g.load_state_dict(model.optimizer_g.state_dict()) if False else None
    _step(model, it + 1, data)
    _step(ref_model, 1, data)
    assert abs(model.get_current_log()['l_pix'] - ref_model.get_current_log()['l_pix']) < 1e-6


def test_forward_overflow_switches_off_the_fp16_split_and_protects_the_weights(monkeypatch, hx2_mode):
    """inputs scaled to 1e6: the first convolutions see operands beyond 65504.  The survey step must not move the weights
    (step guard: non-finite norm), must take the run off the fp16 split, and the next step must be finite."""
    model, K = _model(monkeypatch)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=5)
    data = {'lq': lq * 1e6, 'gt': gt * 1e6, 'ref': ref * 1e6}
    w0 = copy.deepcopy({k: v.detach().clone() for k, v in model.net_g.state_dict().items()})
    try:
        _step(model, 1, data)
        r = model.last_range_survey
        assert r['fwd'][2] > 0 or r['fwd'][1] > model.FWD_MAX_EXP
        assert K.MATH == 'bx3'
        gd = model.optimizer_g.guard.read()
        if gd.skipped:                                                # the overflowing step itself: skipped, weights untouched
            assert all(torch.equal(w0[k], v) for k, v in model.net_g.state_dict().items())
        _step(model, 2, data)
        loss = model.get_current_log()['l_pix']
        assert np.isfinite(loss)
        assert model.optimizer_g.guard.read().finite == 1
    finally:
        K.set_math('hx2')


def test_unhandled_non_finite_loss_is_loud(monkeypatch, hx2_mode):
    """with the survey disabled the same overflow must surface as an exception when the log is read"""
    model, K = _model(monkeypatch)
    monkeypatch.setenv('TDR_RANGE_CHECK_EVERY', '0')
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=5)
    _step(model, 1, {'lq': lq * 1e6, 'gt': gt * 1e6, 'ref': ref * 1e6})
    with pytest.raises(FloatingPointError):
        model.get_current_log()
    assert model.optimizer_g.guard.read().skipped == 1                # and the weights were not touched

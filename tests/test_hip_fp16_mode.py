"""TDR_MATH=h1: plain fp16 MFMA (one product, fp32 accumulation) -- the arithmetic BASELINE configs[4] names ("fp16 MFMA",
SURVEY 8d cfg5).  Reduced precision by construction, so the checks are the ones that configuration is judged on: the
restored image agrees with the fp32-equivalent path to a PSNR far above any restoration PSNR, the gradients point the same
way, and a short training run follows the same loss curve."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    mse = torch.mean((a - b) ** 2).item()
    return 10 * math.log10(1.0 / max(mse, 1e-20))


def _model(arch, math_mode, seed=0):
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.models import create_model
    import bench
    K.set_math(math_mode)
    torch.manual_seed(seed)
    if arch == 'nafnet':
        opt = bench.make_opt(16, [1, 1, 1, 2], 128, False, 'nafnet')
    else:
        opt = bench.make_opt(32, [1, 1, 1, 28], 128, False, 'restormer')
        opt['network_g'].update(dim=16, nf=16, num_blocks=[1, 1, 1, 2], num_refinement_blocks=1)
    m = create_model(opt)
    from textualdegremoval_amd.utils.synthetic import randomize_gates
    randomize_gates(m.net_g)
    return m


@pytest.mark.parametrize('arch', ['nafnet', 'restormer'])
def test_h1_forward_psnr_and_gradient_direction(arch):
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.utils.synthetic import synthetic_pair
    data = {k: v.cuda() for k, v in synthetic_pair(2, 128, 128, seed=7).items()}
    outs, grads = {}, {}
    prev = K.MATH
    try:
        for mode in ('hx2', 'h1'):
            m = _model(arch, mode)
            m.feed_train_data(data)
            m.update_learning_rate(1, warmup_iter=-1)
            m.optimize_parameters(1)
            outs[mode] = m.output.detach().float().clone()
            grads[mode] = torch.cat([q.grad.flatten().double() for q in m.net_g.parameters() if q.grad is not None])
    finally:
        K.set_math(prev)
    assert torch.isfinite(outs['h1']).all()
    p = _psnr(outs['h1'].clamp(0, 1), outs['hx2'].clamp(0, 1))
    assert p > 55.0, f'fp16-MFMA output vs fp32-equivalent output: {p:.1f} dB'
    if True:
        cos = torch.nn.functional.cosine_similarity(grads['h1'], grads['hx2'], dim=0).item()
        assert cos > 0.999, cos


def test_h1_short_training_run_tracks_the_fp32_equivalent_curve():
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd.utils.synthetic import synthetic_pair
    data = {k: v.cuda() for k, v in synthetic_pair(2, 128, 128, seed=11).items()}
    curves = {}
    prev = K.MATH
    try:
        for mode in ('hx2', 'h1'):
            m = _model('nafnet', mode)
            losses = []
            for it in range(1, 13):
                m.update_learning_rate(it, warmup_iter=-1)
                m.feed_train_data(data)
                m.optimize_parameters(it)
                losses.append(float(m.get_current_log()['l_pix']))
            curves[mode] = losses
    finally:
        K.set_math(prev)
    a, b = curves['hx2'], curves['h1']
    assert b[-1] < b[0]
    for x, y in zip(a, b):
        assert abs(x - y) <= 2e-2 * abs(x) + 1e-5, (a, b)

"""Magnitude survey of the gradient operands of the matrix-core kernels in one cfg2 train step (diagnostic; torch ops
are used here only to take statistics).  For every data-gradient convolution and weight-gradient call: max|g| of the
gradient operand and how much of sum|g| lies below max * 2^-12 -- the numbers that decide what pre-scaling the 2-way
fp16 split would need for the backward contractions.   usage: python profiles/grad_range_survey.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TDR_GRAPH'] = '0'
import torch  # noqa: E402

import bench  # noqa: E402
from textualdegremoval_amd import kernels as K  # noqa: E402
from textualdegremoval_amd.models import create_model  # noqa: E402
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair  # noqa: E402

torch.manual_seed(0)
model = create_model(bench.make_opt(32, [1, 1, 1, 28], 512, False))
randomize_gates(model.net_g)
data = {k: v.cuda() for k, v in synthetic_pair(4, 512, 512, seed=1234).items()}


def step(it):
    model.update_learning_rate(it, warmup_iter=-1)
    model.feed_train_data(data)
    model.optimize_parameters(it)


step(1); step(2)
rows = []


def stat(kind, t):
    a = t.detach().abs().float()
    mx = a.max().item()
    tot = a.sum().item()
    small = a[a < mx * 2.0 ** -12].sum().item() if mx > 0 else 0.0
    rows.append((kind, tuple(t.shape), mx, small / max(tot, 1e-30)))


oc, ow = K.conv_forward, K.conv_wgrad
in_bwd = [False]


def conv(x, wp, *a, **kw):
    if in_bwd[0] and getattr(wp, 'fmt', 0) == K.FMT_BX3:
        stat('dgrad', x)
    return oc(x, wp, *a, **kw)


def wgrad(x, dout, *a, **kw):
    stat('wgrad', dout)
    return ow(x, dout, *a, **kw)


ol = K.l1_loss


def l1(*a, **kw):
    r = ol(*a, **kw)
    in_bwd[0] = True
    return r


K.conv_forward, K.conv_wgrad, K.l1_loss = conv, wgrad, l1
step(3)
torch.cuda.synchronize()
mx = [r[2] for r in rows if r[2] > 0]
print(f'{len(rows)} gradient operands; max|g| ranges {min(mx):.3e} .. {max(mx):.3e}  (2^{math.log2(min(mx)):.1f} .. 2^{math.log2(max(mx)):.1f})')
print(f'share of sum|g| below max*2^-12: median {sorted(r[3] for r in rows)[len(rows) // 2]:.2e}, worst {max(r[3] for r in rows):.2e}')
for k in ('dgrad', 'wgrad'):
    rr = [r for r in rows if r[0] == k]
    print(k, len(rr), 'calls; log2(max|g|) histogram:')
    hist = {}
    for r in rr:
        if r[2] > 0:
            b = int(math.floor(math.log2(r[2])))
            hist[b] = hist.get(b, 0) + 1
    print('   ', ' '.join(f'2^{b}:{n}' for b, n in sorted(hist.items())))
worst = sorted(rows, key=lambda r: -r[3])[:5]
for r in worst:
    print('   worst small-mass:', r)

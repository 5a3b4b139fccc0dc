#!/usr/bin/env python
"""Where the ATen fill launches of a step come from: torch.zeros / zeros_like / new_zeros / zero_ / fill_ call sites during one eager step of the
headline workload (counts per repo call site).  usage: python profiles/probe_fills.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

sites = collections.Counter()
on = [False]


def wrap(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        if on[0]:
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ROOT in fr.filename and 'probe_fills' not in fr.filename:
                    sites[f'{name} {os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.line[:90]}'] += 1
                    break
        return f(*a, **k)
    setattr(obj, name, g)


for n in ('zeros', 'zeros_like', 'full', 'ones', 'ones_like'):
    wrap(torch, n)
for n in ('zero_', 'fill_', 'new_zeros', 'new_full'):
    wrap(torch.Tensor, n)

os.environ['TDR_GRAPH'] = '0'                              # eager steps: the call sites are what a capture would record
from textualdegremoval_amd.models import create_model
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair
torch.manual_seed(0)
opt = bench.make_opt(32, [1, 1, 1, 28], 512, False, 'nafnet')
model = create_model(opt)
randomize_gates(model.net_g)
data = {k: v.cuda() for k, v in synthetic_pair(4, 512, 512, seed=1234).items()}
for i in range(4):
    on[0] = i == 3
    model.update_learning_rate(i + 1, warmup_iter=-1)
    model.feed_train_data(data)
    model.optimize_parameters(i + 1)
torch.cuda.synchronize()
for s, c in sites.most_common():
    print(f'{c:4d}  {s}')
print('total', sum(sites.values()))

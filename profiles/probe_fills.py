"""Where do the ATen fill / copy launches of one training step come from?  One eager step of the headline workload under torch.profiler with
Python stacks; aggregates aten::fill_ / zero_ / copy_ / zeros calls that launch a device kernel by their innermost package frame.
    TDR_GRAPH=0 python profiles/probe_fills.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TDR_GRAPH', '0')
import bench  # noqa: E402
from textualdegremoval_amd.models import create_model  # noqa: E402
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair  # noqa: E402

torch.manual_seed(0)
opt = bench.make_opt(32, [1, 1, 1, 28], 512, False, 'nafnet')
model = create_model(opt)
randomize_gates(model.net_g)
data = {k: v.cuda() for k, v in synthetic_pair(4, 512, 512, seed=1234).items()}


def step(it):
    model.update_learning_rate(it, warmup_iter=-1)
    model.feed_train_data(data)
    model.optimize_parameters(it)


for it in range(1, 4):
    step(it)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(4)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::zeros_like', 'aten::clone', 'aten::contiguous', 'aten::to', 'aten::add_', 'aten::mul_', 'aten::add', 'aten::mul', 'aten::sum', 'aten::cat', 'aten::index_select', 'aten::stack') and ev.cpu_parent is None or \
            (ev.name in ('aten::fill_', 'aten::copy_') and ev.cpu_parent is not None and not ev.cpu_parent.name.startswith('aten::')):
        frames = [f for f in (ev.stack or []) if 'textualdegremoval_amd' in f or 'bench.py' in f]
        agg[(ev.name, frames[0] if frames else (ev.stack[0] if ev.stack else '?'))] += 1
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print(f'{n:5d}  {name:18s} {where}')
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith('CUDA'):
        kern[ev.name[:90]] += 1
print('--- device kernels of the step that are not the library\'s own')
for k, n in kern.most_common():
    if 'at::' in k or 'Memcpy' in k or 'Memset' in k or 'copyBuffer' in k or 'fill' in k.lower():
        print(f'{n:5d}  {k}')

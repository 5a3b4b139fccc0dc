"""Tile-configuration sweep of the split-bf16 forward/data-gradient conv kernel over every distinct conv call of one
cfg2 train step: records the calls of an eager step, replays each under every forced configuration
(tdr_conv_force_cfg) and prints the table the selection heuristic in csrc/tdr_conv_bx3.hip (launch_bx_shape) is
checked against.   usage: python profiles/autotune_conv.py [nafnet|restormer]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TDR_GRAPH'] = '0'
import torch  # noqa: E402

import bench  # noqa: E402
from textualdegremoval_amd import _lib, kernels as K  # noqa: E402
from textualdegremoval_amd.models import create_model  # noqa: E402
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else 'nafnet'
CFG1 = {1: '128x128', 2: '64x256', 3: '64x128', 4: '32x256', 5: '256x64'}
CFG3 = {1: '128x256', 2: '64x256', 3: '64x128', 4: '128x128'}
torch.manual_seed(0)
size, batch = (256, 8) if arch == 'restormer' else (512, 4)
model = create_model(bench.make_opt(32, [1, 1, 1, 28], size, False, arch))
randomize_gates(model.net_g)
data = {k: v.cuda() for k, v in synthetic_pair(batch, size, size, seed=1234).items()}


def step(it):
    model.update_learning_rate(it, warmup_iter=-1)
    model.feed_train_data(data)
    model.optimize_parameters(it)


step(1); step(2)
calls = {}
orig = K.conv_forward


def rec(x, wp, Mpad, Cout, KH, stride=1, dil=1, pad=0, **kw):
    out = orig(x, wp, Mpad, Cout, KH, stride=stride, dil=dil, pad=pad, **kw)
    if getattr(wp, 'fmt', 0) in (K.FMT_BX3, K.FMT_HX2) and stride == 1 and Cout > 32:
        key = (KH, x.shape[0], kw.get('Cin') or (x.shape[1] // 2 if kw.get('gate') else x.shape[1]), Cout, x.shape[2], x.shape[3],
               kw.get('epi', 0), bool(kw.get('gate')), 'hx2' if wp.fmt == K.FMT_HX2 else 'bx3')
        if key not in calls:
            kw2 = dict(kw); kw2['out'] = out
            calls[key] = [0, (x, wp, Mpad, Cout, KH, stride, dil, pad, kw2)]
        calls[key][0] += 1
    return out


K.conv_forward = rec
step(3)
K.conv_forward = orig
torch.cuda.synchronize()
lib = _lib.load()


def timeit(args, iters=10):
    x, wp, Mpad, Cout, KH, stride, dil, pad, kw = args
    for _ in range(2):
        orig(x, wp, Mpad, Cout, KH, stride=stride, dil=dil, pad=pad, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        orig(x, wp, Mpad, Cout, KH, stride=stride, dil=dil, pad=pad, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot_def = tot_best = 0.0
print(f'{"KH N Cin->Cout @HxW epi gate":40s} {"n/step":>6s} {"default":>9s}  forced configs (us)')
for key, (cnt, args) in sorted(calls.items(), key=lambda kv: -kv[1][0]):
    KH = key[0]
    lib.tdr_conv_force_cfg(KH, 0)
    base = timeit(args)
    row, best, bestc = [], base, 'default'
    for c, name in (CFG1 if KH == 1 else CFG3).items():
        lib.tdr_conv_force_cfg(KH, c)
        try:
            t = timeit(args)
        except RuntimeError:
            t = float('nan')
        row.append(f'{name}:{t:7.1f}')
        if t < best * 0.97:
            best, bestc = t, name
    lib.tdr_conv_force_cfg(KH, 0)
    tot_def += cnt * base
    tot_best += cnt * best
    tag = f'{key[0]}x{key[0]} N{key[1]} {key[2]}->{key[3]} @{key[4]}x{key[5]} e{key[6]} g{int(key[7])} {key[8]}'
    print(f'{tag:44s} {cnt:6d} {base:9.1f}  {"  ".join(row)}   best {bestc}')
print(f'sum over the step: heuristic {tot_def / 1e3:.2f} ms, per-shape best {tot_best / 1e3:.2f} ms')

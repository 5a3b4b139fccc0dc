"""The 64x64-level NAFBlock chain of BASELINE configs[1] (c = 256, N = 4, 28 blocks with their own weights) as captured graphs:
forward only, forward + data-gradient chain without the leaf weight gradients (ablation), and everything -- where the 12 ms of this
level go.  python profiles/probe_naf_chain.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from textualdegremoval_amd import engine as E, kernels as K  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_hip_nafblock_fused import block_params, rnd  # noqa: E402

N, c, H, W = 4, 256, 64, 64
NB = int(os.environ.get('NB', '28'))
x = rnd(N, c, H, W, seed=1).cuda()
dout = rnd(N, c, H, W, seed=2).cuda()
Ps = [{k: v.cuda() for k, v in block_params(c, 3 + i).items()} for i in range(NB)]
plan = K.PackPlan()
K.set_pack_plan(plan)


def fwd():
    h, saved = x, []
    for P in Ps:
        h, sv = E.naf_fwd(h, P)
        saved.append(sv)
    return h, saved


def bwd(saved, leaves):
    d = dout
    G = {}
    if leaves == 'late':
        with E.deferred_join(), E.late_leaves(G):
            for i in reversed(range(NB)):
                E.set_late_prefix(f'b{i}.')
                d, g = E.naf_bwd(d, Ps[i], saved[i])
            E.run_late_leaves(G, lambda: None)
    else:
        with E.deferred_join():
            for P, sv in zip(reversed(Ps), reversed(saved)):
                d, g = E.naf_bwd(d, P, sv)
    return d


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    plan.run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    refs = []
    with K.workspace_capture(refs), torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps):
        g.replay()
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps / NB * 1e3, (g, refs)


t_f, keep0 = timed(lambda: fwd())
_, saved = fwd()
for rep in range(2):
    t_full, keep1 = timed(lambda: bwd(saved, 'side'))
    t_late, keep2 = timed(lambda: bwd(saved, 'late'))
    orig = E._leaf_wgrad1x1
    E._leaf_wgrad1x1 = lambda *a, **k: None
    t_nolf, keep3 = timed(lambda: bwd(saved, 'side'))
    E._leaf_wgrad1x1 = orig
    print(f'per block: forward {t_f:.1f} us; backward with leaves on the side stream {t_full:.1f}, leaves deferred + grouped '
          f'{t_late:.1f}, without the conv1/4/5 weight gradients {t_nolf:.1f} us', flush=True)
    del keep1, keep2, keep3

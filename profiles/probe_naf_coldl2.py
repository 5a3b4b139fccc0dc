"""Does the 64x64-level NAFBlock chain pay for COLD weights?  28 blocks forward + backward as one captured graph, (a) every block with
its own weights (the step's situation: 2.4 MB of packed planes per block, packed once at step start, 28 blocks = 67 MB > the 32 MB of
L2), (b) all blocks sharing ONE set of weights (L2-warm after the first block).  python profiles/probe_naf_coldl2.py"""
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


def chain(Ps):
    h, saved = x, []
    for P in Ps:
        h, sv = E.naf_fwd(h, P)
        saved.append(sv)
    d = dout
    with E.deferred_join():
        for P, sv in zip(reversed(Ps), reversed(saved)):
            d, G = E.naf_bwd(d, P, sv)
    return h, d


for name, distinct in (('distinct weights', True), ('shared weights', False), ('distinct weights', True), ('shared weights', False)):
    plan = K.PackPlan()
    K.set_pack_plan(plan)
    if distinct:
        Ps = [{k: v.cuda() for k, v in block_params(c, 3 + i).items()} for i in range(NB)]
    else:
        P0 = {k: v.cuda() for k, v in block_params(c, 3).items()}
        Ps = [P0] * NB
    for rep in range(2):
        chain(Ps)
    plan.run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    refs = []
    with K.workspace_capture(refs), torch.cuda.graph(g):
        chain(Ps)
    for rep in range(3):
        g.replay()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for rep in range(10):
        g.replay()
    e[1].record()
    torch.cuda.synchronize()
    print(f'{name}: {e[0].elapsed_time(e[1]) / 10 / NB * 1e3:.1f} us per block (fwd + bwd, {NB} blocks)', flush=True)
    del g, refs

"""The grouped 1x1 weight-gradient launch (tdr_wgrad1x1_group) alone at the C = 256 level of configs[1]: NP problems of 256 -> 512 @ 64^2, N = 4 (cold
operands: NP x 50 MB), time per launch and -- under rocprofv3 --pmc -- its L2 hit rate / HBM fetch against the 50 MB per problem it needs.
usage: python profiles/probe_wgrad1x1_group.py [NP]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textualdegremoval_amd import kernels as K
K.set_math('bx3')
torch.manual_seed(0)
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 58
reqs = [(torch.randn(4, 256, 64, 64, device='cuda'), torch.randn(4, 512, 64, 64, device='cuda') * 1e-3, 512, 256, False) for _ in range(NP)]
for _ in range(2):
    out = K.wgrad1x1_group(reqs, seq=700)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    out = K.wgrad1x1_group(reqs, seq=700)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 3 * 1e3
flop = NP * 2.0 * 4 * 4096 * 512 * 256
print(f'grouped {NP} x (256 -> 512 @64^2 N4): {us:8.1f} us ({flop / us * 1e-6:5.0f} TF), algorithmic bytes {NP * 50.3:.0f} MB', flush=True)

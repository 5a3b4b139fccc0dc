"""The un-guided SFNet (round 6, sfnet_engine.py) at the reference's default depth: forward + backward time and the kernels it spends it in.
python profiles/probe_sfnet.py [N H W num_res]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import sfnet_oracle as SO  # noqa: E402   (seeded state only: this is a probe, not the product path)
from textualdegremoval_amd import kernels as K, sfnet_engine as SE  # noqa: E402

N, H, W, num_res = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (4, 256, 256, 16)))
sd = SO.synth_state(num_res, seed=1)
for k in sd:
    if k.endswith('conv2.main.0.weight'):
        sd[k] = sd[k] * 0.2
P = {k: v.cuda() for k, v in sd.items()}
x = torch.rand(N, 3, H, W).cuda()
for _ in range(2):
    outs, saved = SE.net_fwd(P, x, num_res)
    G = SE.net_bwd([torch.randn_like(o) / o.numel() for o in outs], P, saved)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
gos = [torch.randn_like(o) / o.numel() for o in outs]
e[0].record()
outs, saved = SE.net_fwd(P, x, num_res)
e[1].record()
G = SE.net_bwd(gos, P, saved)
e[2].record()
torch.cuda.synchronize()
print(f'SFNet num_res {num_res}, {N} x {H}x{W} [{K.MATH}], eager: forward {e[0].elapsed_time(e[1]):.1f} ms, backward {e[1].elapsed_time(e[2]):.1f} ms '
      f'({N / (e[0].elapsed_time(e[2]) * 1e-3):.1f} img/s); peak memory {torch.cuda.max_memory_allocated() / 1e9:.1f} GB')

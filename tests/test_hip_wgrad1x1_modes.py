"""The 1x1 weight-gradient kernels of csrc/tdr_wgrad_1x1.hip (split-once on 4 waves for 128 x 128 tiles, the LDS-DMA ring for 64 x 64
tiles) against float64 on the NAFBlock leaf shapes: odd and even stage counts, ragged channel tiles, the SimpleGate operand, the fused
bias gradient, per-image groups, W < 8.  Runs in a child process with TDR_MATH=bx3.  (The kernel selector TDR_WG1_SP -- 0: LDS-DMA ring
everywhere, 1: split-once on 8 waves with the in-block K split -- exists only in tuning builds of the library since round 6:
`make -C textualdegremoval_amd/csrc variant VFILE=tdr_wgrad_1x1 VFLAGS=-DTDR_TUNING_KNOBS`; with such a build loaded through
TDR_LIB_PATH the parametrisation below covers all three.)  Replaces autograd's weight gradients of the reference's 1x1 convolutions
(models/archs/network_nafnet_guided_arch.py:183-205,216-238)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from textualdegremoval_amd import kernels as K
torch.manual_seed(0)
worst = 0.0
# (N, Cin, Cout, H, W, gate, per_image): 32-pixel stages per image = H * W / 32
for (N, Cin, Cout, H, W, gate, pi) in [(2, 128, 128, 32, 32, False, False), (1, 96, 72, 40, 40, False, False), (3, 128, 256, 24, 12, False, False),
                                       (2, 256, 512, 8, 12, False, False), (2, 128, 128, 16, 18, True, False), (3, 96, 80, 32, 40, True, True),
                                       (2, 160, 136, 8, 20, False, True), (1, 128, 128, 8, 4, False, False),
                                       (2, 64, 64, 8, 4, False, True), (3, 128, 96, 16, 4, True, True)]:   # per-image with W < 8 (ADVICE r5)
    x = torch.randn(N, Cin * (2 if gate else 1), H, W, device='cuda') * 3e-4       # gradient-sized magnitudes: no fp16 window here
    d = torch.randn(N, Cout, H, W, device='cuda') * 2e-5
    g, db = K.conv_wgrad(x, d, Cout, Cin, 1, gate=gate, per_image=pi, want_db=True)
    xe = (x[:, :Cin] * x[:, Cin:]) if gate else x
    ref = torch.einsum('nkp,ncp->nkc', d.double().flatten(2), xe.double().flatten(2))
    if not pi:
        ref = ref.sum(0, keepdim=True)
    e = (g.double().view_as(ref) - ref).abs().max().item() / ref.abs().max().item()
    rb = d.double().sum((0, 2, 3))
    eb = (db.double() - rb).abs().max().item() / rb.abs().max().item()
    worst = max(worst, e, eb)
    assert e < 2e-6 and eb < 2e-6, (N, Cin, Cout, H, W, gate, pi, e, eb)
print('WORST', worst)
'''


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['0', '1', '2'] if 'tuning' in os.environ.get('TDR_LIB_PATH', '') else ['2'])
def test_wgrad1x1_kernel_modes_vs_fp64(mode):
    env = dict(os.environ, TDR_WG1_SP=mode, TDR_MATH='bx3')
    out = subprocess.run([sys.executable, '-c', CHILD % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'WORST' in out.stdout

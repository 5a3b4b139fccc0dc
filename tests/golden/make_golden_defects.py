"""Record, by running the REFERENCE, behaviour this repo mirrors as an error rather than as a feature.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_defects.py
Writes tests/golden/reference_defects.npz (strings only).

R7 -- NAFNetLocal_RefFusion (models/archs/network_nafnet_guided_arch.py:743-753), the TLSC test-time wrapper of the guided
NAFNet, cannot be constructed: Local_Base.convert (nafnet_local_arch.py:99-104) calls `self.forward(imgs)` with one argument
and NAFNetRefFusion.forward needs (inp, ref).  No YAML names the class.

R8 -- SFNetRefFusion (models/archs/network_sfnet_guided_arch.py:410-797) constructs but cannot run a forward pass for any
width: its MASA Encoder (:292-317) builds conv_L2 / conv_L3 with nf output channels and feeds them to residual blocks of
2nf / 4nf channels (RuntimeError in the first ResidualBlock of blk_L2), returns three feature levels where forward reads
feat[4] (:621), and EBlockResFusion.forward (:180-186) multiplies the un-called nn.Sequential by alpha.  The YAML
006_sfnet_image_dehazing_outdoor.yml therefore cannot train; there is no behaviour to be in parity with."""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, REF)
m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
mod = importlib.import_module('models.archs.network_nafnet_guided_arch')
try:
    mod.NAFNetLocal_RefFusion(width=8, nf=8, enc_blk_nums=[1] * 4, dec_blk_nums=[1] * 4, middle_blk_num=1, ext_n_blocks=[1] * 4,
                              reffusion_n_blocks=[1] * 5, train_size=(1, 3, 64, 64))
    r7 = 'constructed'
except Exception as e:  # noqa: BLE001
    r7 = f'{type(e).__name__}: {e}'
import torch  # noqa: E402

sf = importlib.import_module('models.archs.network_sfnet_guided_arch')
r8 = []
for nf in (32, 64):
    try:
        net = sf.SFNetRefFusion(mode='train', num_res=2, nf=nf, ext_n_blocks=[1] * 4, reffusion_n_blocks=[2] * 4)
        with torch.no_grad():
            net(torch.rand(1, 3, 64, 64), torch.rand(1, 3, 64, 64))
        r8.append('ran')
    except Exception as e:  # noqa: BLE001
        r8.append(f'{type(e).__name__}: {e}')
np.savez(os.path.join(HERE, 'reference_defects.npz'), r7_nafnetlocal_reffusion=np.array(r7), r8_sfnet_reffusion=np.array(r8))
print(r7)
print(r8)

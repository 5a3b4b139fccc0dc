"""Record, by running the REFERENCE, behaviour this repo mirrors as an error rather than as a feature.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_defects.py
Writes tests/golden/reference_defects.npz (strings only).

R7 -- NAFNetLocal_RefFusion (models/archs/network_nafnet_guided_arch.py:743-753), the TLSC test-time wrapper of the guided
NAFNet, cannot be constructed: Local_Base.convert (nafnet_local_arch.py:99-104) calls `self.forward(imgs)` with one argument
and NAFNetRefFusion.forward needs (inp, ref).  No YAML names the class."""
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
np.savez(os.path.join(HERE, 'reference_defects.npz'), r7_nafnetlocal_reffusion=np.array(r7))
print(r7)

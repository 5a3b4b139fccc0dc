"""Workload for the PMC (HBM traffic) passes: calibration copies of known size, then one eager train step.
Run under rocprofv3 once per counter (the TCC block cannot hold FETCH_SIZE and WRITE_SIZE together):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_fetch -- python profiles/pmc_workload.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_write -- python profiles/pmc_workload.py
then  python profiles/pmc_summarize.py out_fetch out_write > profiles/r1/pmc_traffic.json
Calibration (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 and
other widths / WRITE_SIZE are uncalibrated, so the factors are measured here on copies of known byte counts with the
same access widths our kernels use (float4 rows_kernel, dword rows_scalar_kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TDR_GRAPH'] = '0'
import torch  # noqa: E402

import bench  # noqa: E402
from textualdegremoval_amd import kernels as K  # noqa: E402
from textualdegremoval_amd.models import create_model  # noqa: E402
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair  # noqa: E402

torch.cuda.set_device(0)
# ---- calibration: 512 MiB copies (larger than the 256 MiB Infinity Cache), float4 path and dword path
n = 128 * 1024 * 1024
src = torch.randn(n, device='cuda')
dst = torch.empty(n, device='cuda')
for _ in range(2):
    K.copy_rows(src, 0, dst, 0, 1, n)                      # rows_kernel<false>: 16 B per lane
    K.copy_rows(src[1:], 0, dst[1:], 0, 1, n - 4)          # misaligned -> rows_scalar_kernel<false>: 4 B per lane
torch.cuda.synchronize()
del src, dst
# ---- one eager step (two warm-up steps first so allocations / packing plans exist)
torch.manual_seed(0)
model = create_model(bench.make_opt(32, [1, 1, 1, 28], 512, False))
randomize_gates(model.net_g)
data = {k: v.cuda() for k, v in synthetic_pair(4, 512, 512, seed=1234).items()}
for it in range(1, 4):
    model.update_learning_rate(it, warmup_iter=-1)
    model.feed_train_data(data)
    model.optimize_parameters(it)
torch.cuda.synchronize()
print('pmc workload done')

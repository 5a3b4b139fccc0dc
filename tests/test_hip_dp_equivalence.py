"""Data parallelism is an identity on the math: 2 ranks x 1 sample (rank-strided shards, gradient all-reduce (mean),
redundant clip + AdamW, loss reduced to rank 0) must reproduce 1 rank x 2 samples -- same per-step rank-0 `l_pix`, same
parameters after 4 steps (2 eager, hipGraph capture, replay) -- although every rank seeds its RNG differently
(manual_seed + rank, main_train_restoration_with_ref_input.py:55) and only the rank-0 parameter broadcast that takes
the DistributedDataParallel constructor's place (models/base_model.py:76-82) makes the replicas start equal."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, 'tests', '_dp_equivalence.py')


def _launch(nproc, out_path, port=29561):
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ['-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
                '--master-port', str(port)]
    cmd += [HELPER, out_path]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and 'DP_EQUIV_DONE' in out.stdout, (out.stdout[-1500:], out.stderr[-2500:])
    return torch.load(out_path)


def test_two_ranks_times_one_sample_equals_one_rank_times_two(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    one = _launch(1, str(tmp_path / 'one.pt'))
    two = _launch(2, str(tmp_path / 'two.pt'))
    for a, b in zip(one['losses'], two['losses']):
        assert abs(a - b) < 2e-7, (one['losses'], two['losses'])
    worst = max((one['params'][k] - two['params'][k]).abs().max().item() for k in one['params'])
    moved = max((one['params'][k]).abs().max().item() for k in one['params'])
    assert worst <= 1e-6, worst
    assert moved > 0

import os
import sys

import pytest
import torch

# The CPU oracle (torch fp32 on the host) runs inside both test tiers.  On the 256-CPU GPU hosts an unbounded intra-op
# pool makes the MKLDNN convolutions collapse and starves the HIP runtime's own threads (a D2H copy then waits for
# minutes), so the pool is capped for the whole session -- test modules must not raise it again.
torch.set_num_threads(min(16, os.cpu_count() or 1))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture
def hx2_mode():
    """the opt-in fast arithmetic (TDR_MATH=hx2: fp16 pair planes, loss-scaled backward, step guard, range survey) for the tests of
    exactly that machinery; the library default is the reference's arithmetic (bx3)"""
    from textualdegremoval_amd import kernels as K
    prev = K.MATH
    K.set_math('hx2')
    yield K
    K.set_math(prev)

"""helper of tests/test_hip_dp_smoke.py: a 1-rank RCCL process group driving 5 train steps (2 eager with bucketed
all-reduce, capture, 2 replays with the flat all-reduce) and comparing the losses with a non-distributed model."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from textualdegremoval_amd.models import create_model  # noqa: E402
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair  # noqa: E402


def run(dist_on):
    torch.manual_seed(0)
    model = create_model(bench.make_opt(8, [1, 1, 1, 1], 128, dist_on))
    randomize_gates(model.net_g)
    data = {k: v.cuda() for k, v in synthetic_pair(1, 128, 128, seed=3).items()}
    losses = []
    for it in range(1, 6):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data(data)
        model.optimize_parameters(it)
        losses.append(model.get_current_log()['l_pix'])
    return losses


if __name__ == '__main__':
    torch.cuda.set_device(0)
    dist.init_process_group('nccl')
    assert dist.get_world_size() == 1
    a = run(True)
    b = run(False)
    dist.destroy_process_group()
    assert all(abs(x - y) < 1e-7 for x, y in zip(a, b)), (a, b)
    print('RCCL_SINGLE_RANK_OK', a)

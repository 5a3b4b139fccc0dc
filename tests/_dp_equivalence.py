"""helper of tests/test_hip_dp_equivalence.py.  Launched (a) by torch.distributed.run with 2 ranks sharing the test
box's one GPU (gloo transport) and (b) as a single process: both train the same network for 4 steps (2 eager, capture,
replay) on the same global batch of 2 samples -- rank r of (a) holds sample r, as EnlargedSampler would deal them --
and write losses + final parameters to argv[1].  Rank r seeds its RNG with manual_seed + r like the reference trainer
(main_train_restoration_with_ref_input.py:55): only the rank-0 broadcast of model_to_device makes the replicas equal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from textualdegremoval_amd.data.data_sampler import EnlargedSampler  # noqa: E402
from textualdegremoval_amd.models import create_model  # noqa: E402
from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair  # noqa: E402

if __name__ == '__main__':
    out_path = sys.argv[1]
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo')
    torch.manual_seed(100 + rank)
    # 0.25 MiB buckets: several buckets even for this small network, so the captured step of the 2-rank run is cut into
    # several graph segments with a bucket exchange between them (image_restoration_ref_model._graph_step)
    model = create_model(bench.make_opt(8, [1, 1, 1, 1], 128, world > 1, bucket_mb=0.25))
    if rank == 0:
        randomize_gates(model.net_g, seed=5)       # a rank-0-only change BEFORE the sync below would be lost; after it, ranks differ ->
    if world > 1:
        model.sync_from_rank0(model.net_g)         # -> so re-sync, as load_network does after a checkpoint load
    full = synthetic_pair(2, 128, 128, seed=77)
    sampler = EnlargedSampler(list(range(2)), world, rank, 1)
    losses = []
    for it in range(1, 5):
        sampler.set_epoch(0)
        idx = sorted(sampler.indices())            # world 1: both samples; world 2: one each
        data = {k: v[idx].cuda() for k, v in full.items()}
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data(data)
        model.optimize_parameters(it)
        losses.append(model.get_current_log()['l_pix'])
    if world > 1:
        st, red = model._gstate, model.grad_reducer
        assert st['split'] and len(st['segs']) == len(red.buckets) >= 3, (len(st['segs']), len(red.buckets))
        assert red.bucket_launches >= 2 * len(red.buckets)
    if rank == 0:
        torch.save({'losses': losses, 'params': {k: v.detach().cpu() for k, v in model.net_g.state_dict().items()}}, out_path)
    if world > 1:
        # every rank must hold the same replica
        flat = torch.cat([p.detach().reshape(-1) for p in model.net_g.parameters()]).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), 'replicas diverged'
        dist.barrier()
        dist.destroy_process_group()
    print('DP_EQUIV_DONE', rank, losses)

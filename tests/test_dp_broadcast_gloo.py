"""without DistributedDataParallel nothing made the replicas start equal.  Two gloo ranks on the CPU
seed differently (manual_seed + rank, like the reference trainer :55), build the model in distributed mode and must
hold identical parameters and buffers afterwards; a non-strict partial checkpoint load re-syncs as well."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _opt(tmp):
    return {'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 0, 'dist': True, 'is_train': True,
            'network_g': dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=[1, 1, 1, 1], dec_blk_nums=[1, 1, 1, 1],
                              middle_blk_num=1, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]),
            'path': {'pretrain_network_g': tmp, 'strict_load_g': False},
            'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                      'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                    'eta_mins': [3e-4, 1e-6]},
                      'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                      'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
            'logger': {'check_freq': 10 ** 9}, 'val': {}, 'scale': 1}


def _worker(rank, world, port, ckpt, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from textualdegremoval_amd.models import create_model
    torch.manual_seed(100 + rank)
    model = create_model(_opt(ckpt))          # model_to_device broadcast, then a partial non-strict load + re-sync
    flat = torch.cat([p.detach().reshape(-1) for p in model.net_g.parameters()])
    q.put((rank, flat.double().sum().item(), flat[::997].tolist()))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_replicas_start_from_rank0_weights(tmp_path):
    # a partial checkpoint: only the intro conv, so everything else keeps its (rank-specific) random init
    ckpt = str(tmp_path / 'partial.pth')
    torch.save({'params': {'intro.weight': torch.full((8, 3, 3, 3), 0.25), 'intro.bias': torch.zeros(8)}}, ckpt)
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, ckpt, q)) for r in range(world)]
    [p.start() for p in ps]
    out = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    [p.join(10) for p in ps]
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    # and they are rank 0's weights: a lone process seeded like rank 0 builds the same tensor
    from textualdegremoval_amd.models.archs import define_network
    torch.manual_seed(100)
    o = _opt(ckpt)
    net = define_network(dict(o['network_g']))
    sd = net.state_dict(); sd.update(torch.load(ckpt)['params']); net.load_state_dict(sd)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert flat[::997].tolist() == out[0][2]

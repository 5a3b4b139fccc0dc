"""Process-group helpers with the reference's names (utils/utils_dist.py:10-83).
backend 'nccl' IS RCCL on ROCm; one process per GPU; env rendezvous
(RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT) kept so
`--launcher pytorch` style launches still work."""
import functools
import os

import torch
import torch.distributed as dist


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    if launcher not in ('pytorch', 'slurm'):
        raise ValueError(f'Invalid launcher type: {launcher}')
    if launcher == 'slurm':
        os.environ.setdefault('RANK', os.environ['SLURM_PROCID'])
        os.environ.setdefault('WORLD_SIZE', os.environ['SLURM_NTASKS'])
        os.environ.setdefault('MASTER_PORT', '29500')
    rank = int(os.environ['RANK'])
    if torch.cuda.is_available():
        local = int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1)))
        torch.cuda.set_device(local)
    elif backend == 'nccl':
        backend = 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        rank, _ = get_dist_info()
        if rank == 0:
            return func(*args, **kwargs)
    return wrapper

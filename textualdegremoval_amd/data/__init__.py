"""`create_dataset` / `create_dataloader` with the reference's contract (data/__init__.py:30-122): datasets are
looked up by class name in the `*_dataset.py` modules of this folder; the train loader takes `batch_size_per_gpu`
(times num_gpu when not distributed), drops the last partial batch, shuffles only without a sampler, seeds worker w
of rank r with `num_workers * r + w + seed`; val/test loaders are batch 1, in order."""
import importlib
import os
import random
from functools import partial

import numpy as np
import torch.utils.data

from ..utils.logger import get_root_logger
from ..utils.utils_dist import get_dist_info
from .prefetch_dataloader import PrefetchDataLoader

__all__ = ['create_dataset', 'create_dataloader']

_here = os.path.dirname(os.path.abspath(__file__))
_dataset_modules = [importlib.import_module(f'{__name__}.{f[:-3]}') for f in sorted(os.listdir(_here)) if f.endswith('_dataset.py')]


def create_dataset(dataset_opt):
    kind = dataset_opt['type']
    for mod in _dataset_modules:
        cls = getattr(mod, kind, None)
        if cls is not None:
            ds = cls(dataset_opt)
            get_root_logger().info(f'Dataset {cls.__name__} - {dataset_opt["name"]} is created.')
            return ds
    raise ValueError(f'Dataset {kind} is not found.')


def _seed_worker(worker_id, num_workers, rank, seed):
    s = num_workers * rank + worker_id + seed
    np.random.seed(s)
    random.seed(s)


def create_dataloader(dataset, dataset_opt, num_gpu=1, dist=False, sampler=None, seed=None):
    phase = dataset_opt['phase']
    rank, _ = get_dist_info()
    if phase == 'train':
        mult = 1 if (dist or num_gpu == 0) else num_gpu
        workers = dataset_opt['num_worker_per_gpu'] * mult
        args = dict(dataset=dataset, batch_size=dataset_opt['batch_size_per_gpu'] * mult, shuffle=sampler is None,
                    num_workers=workers, sampler=sampler, drop_last=True,
                    worker_init_fn=partial(_seed_worker, num_workers=workers, rank=rank, seed=seed) if seed is not None else None)
    elif phase in ('val', 'test'):
        args = dict(dataset=dataset, batch_size=1, shuffle=False, num_workers=0)
    else:
        raise ValueError(f"Wrong dataset phase: {phase}. Supported ones are 'train', 'val' and 'test'.")
    args['pin_memory'] = dataset_opt.get('pin_memory', False)
    if dataset_opt.get('prefetch_mode') == 'cpu':
        depth = dataset_opt.get('num_prefetch_queue', 1)
        get_root_logger().info(f'Use cpu prefetch dataloader: num_prefetch_queue = {depth}')
        return PrefetchDataLoader(num_prefetch_queue=depth, **args)
    return torch.utils.data.DataLoader(**args)       # prefetch_mode None, or 'cuda' (the trainer wraps it in CUDAPrefetcher)

"""Experiment-directory / seeding / resume helpers under the reference's names (utils/utils_misc.py)."""
import os
import random
import time

import numpy as np
import torch

from .logger import get_root_logger
from .utils_dist import master_only


def scandir(dir_path, suffix=None, recursive=False, full_path=False):
    """yield the (relative, or full) paths of the non-hidden files under dir_path, optionally filtered by suffix"""
    if suffix is not None and not isinstance(suffix, (str, tuple)):
        raise TypeError('"suffix" must be a string or tuple of strings')

    def walk(d):
        for e in os.scandir(d):
            if e.is_file() and not e.name.startswith('.'):
                p = e.path if full_path else os.path.relpath(e.path, dir_path)
                if suffix is None or p.endswith(suffix):
                    yield p
            elif recursive and e.is_dir():
                yield from walk(e.path)
    return walk(dir_path)


def get_time_str():
    return time.strftime('%Y%m%d_%H%M%S', time.localtime())


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def mkdir_and_rename(path):
    """a fresh directory at `path`; an existing one is kept under `<path>_archived_<time>`"""
    if os.path.exists(path):
        archived = f'{path}_archived_{get_time_str()}'
        print(f'Path already exists. Rename it to {archived}', flush=True)
        os.rename(path, archived)
    os.makedirs(path, exist_ok=True)


@master_only
def make_exp_dirs(opt):
    paths = dict(opt['path'])
    mkdir_and_rename(paths.pop('experiments_root' if opt['is_train'] else 'results_root'))
    for key, p in paths.items():
        if not any(s in key for s in ('strict_load', 'pretrain_network', 'resume', 'pretrain_dino', 'param_key')) and isinstance(p, str):
            os.makedirs(p, exist_ok=True)


def check_resume(opt, resume_iter):
    """when resuming, every network_* is reloaded from models/net_<x>_<iter>.pth (pretrain paths are overridden)"""
    logger = get_root_logger()
    if not opt['path'].get('resume_state'):
        return
    nets = [k for k in opt if k.startswith('network_')]
    if any(opt['path'].get(f'pretrain_{n}') is not None for n in nets):
        logger.warning('pretrain_network path will be ignored during resuming.')
    ignore = opt['path'].get('ignore_resume_networks')
    for n in nets:
        base = n.replace('network_', '')
        if ignore is None or base not in ignore:
            opt['path'][f'pretrain_{n}'] = os.path.join(opt['path']['models'], f'net_{base}_{resume_iter}.pth')
            logger.info(f"Set pretrain_{n} to {opt['path'][f'pretrain_{n}']}")

"""Logging helpers under the reference's names (utils/logger.py): one 'basicsr' root logger (rank 0 at INFO with an
optional file handler, other ranks ERROR only), the per-`print_freq` message line of the trainer, and the optional
tensorboard / wandb hooks (imported lazily: neither package is needed unless the YAML asks for it)."""
import datetime
import logging
import time

from .utils_dist import get_dist_info, master_only

_configured = set()
_FORMAT = '%(asctime)s %(levelname)s: %(message)s'


def get_root_logger(logger_name='basicsr', log_level=logging.INFO, log_file=None):
    logger = logging.getLogger(logger_name)
    if logger_name in _configured:
        return logger
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter(_FORMAT))
    logger.addHandler(handler)
    logger.propagate = False
    rank, _ = get_dist_info()
    if rank != 0:
        logger.setLevel('ERROR')
    else:
        logger.setLevel(log_level)
        if log_file is not None:
            fh = logging.FileHandler(log_file, 'w')
            fh.setFormatter(logging.Formatter(_FORMAT))
            fh.setLevel(log_level)
            logger.addHandler(fh)
    _configured.add(logger_name)
    return logger


class MessageLogger:
    """callable taking the trainer's log_vars dict {epoch, iter, lrs, [time, data_time], <losses...>}"""

    def __init__(self, opt, start_iter=1, tb_logger=None):
        self.exp_name = opt['name']
        self.interval = opt['logger']['print_freq']
        self.start_iter = start_iter
        self.max_iters = opt['train']['total_iter']
        self.use_tb_logger = opt['logger'].get('use_tb_logger', False)
        self.tb_logger = tb_logger
        self.start_time = time.time()
        self.logger = get_root_logger()

    @master_only
    def __call__(self, log_vars):
        epoch, it, lrs = log_vars.pop('epoch'), log_vars.pop('iter'), log_vars.pop('lrs')
        parts = [f'[{self.exp_name[:5]}..][epoch:{epoch:3d}, iter:{it:8,d}, lr:(' + ''.join(f'{v:.3e},' for v in lrs) + ')] ']
        if 'time' in log_vars:
            iter_time, data_time = log_vars.pop('time'), log_vars.pop('data_time')
            per_iter = (time.time() - self.start_time) / (it - self.start_iter + 1)
            eta = datetime.timedelta(seconds=int(per_iter * (self.max_iters - it - 1)))
            parts.append(f'[eta: {eta}, time (data): {iter_time:.3f} ({data_time:.3f})] ')
        for k, v in log_vars.items():
            parts.append(f'{k}: {v:.4e} ')
            if self.use_tb_logger and self.tb_logger is not None and 'debug' not in self.exp_name:
                self.tb_logger.add_scalar(f'losses/{k}' if k.startswith('l_') else k, v, it)
        self.logger.info(''.join(parts))


@master_only
def init_tb_logger(log_dir):
    from torch.utils.tensorboard import SummaryWriter
    return SummaryWriter(log_dir=log_dir)


@master_only
def init_wandb_logger(opt):
    """wandb only mirrors the tensorboard log"""
    import wandb
    cfg = opt['logger']['wandb']
    rid = cfg.get('resume_id')
    wandb.init(id=rid or wandb.util.generate_id(), resume='allow' if rid else 'never', name=opt['name'], config=opt,
               project=cfg['project'], sync_tensorboard=True)
    logging.getLogger('basicsr').info(f"Use wandb logger; project={cfg['project']}.")


def get_env_info():
    import torch

    from .. import __version__
    hip = getattr(torch.version, 'hip', None)
    return ('\ntextualdegremoval_amd (MI355X-native guided restoration)'
            f'\nVersion Information:\n\ttextualdegremoval_amd: {__version__}\n\tPyTorch: {torch.__version__}\n\tHIP: {hip}')

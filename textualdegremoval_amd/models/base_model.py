"""BaseModel with the reference's method surface (models/base_model.py): device
placement, schedulers, LR warm-up, checkpoints (`{'params': state_dict}` /
`.state` files), loss reduction.  DistributedDataParallel is replaced by the
explicit RCCL gradient all-reduce of textualdegremoval_amd.parallel."""
import logging
import os
from collections import OrderedDict
from copy import deepcopy

import torch

from . import lr_scheduler
from ..parallel import GradAllReducer, reduce_loss_to_rank0
from ..utils.utils_dist import get_dist_info, master_only

logger = logging.getLogger('basicsr')


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device('cuda' if opt['num_gpu'] != 0 else 'cpu')
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []

    # -- hooks overridden by concrete models
    def feed_data(self, data):
        pass

    def optimize_parameters(self, current_iter):
        pass

    def get_current_visuals(self):
        pass

    def save(self, epoch, current_iter):
        pass

    def validation(self, dataloader, current_iter, tb_logger, save_img=False, rgb2bgr=True, use_image=True):
        if self.opt['dist']:
            return self.dist_validation(dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image)
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image)

    def model_ema(self, decay=0.999):
        """p_ema = decay * p_ema + (1 - decay) * p for every parameter (reference :54-62); on the GPU one
        multi-tensor launch (tdr_multi_ema) over a pointer table built once."""
        src = dict(self.get_bare_model(self.net_g).named_parameters())
        pairs = [(src[k].data, p.data) for k, p in self.net_g_ema.named_parameters()]
        if not pairs or not pairs[0][0].is_cuda:
            for a, b in pairs:
                b.mul_(decay).add_(a, alpha=1 - decay)
            return
        from .. import _lib
        from .. import kernels as K
        key = tuple((a.data_ptr(), b.data_ptr()) for a, b in pairs)
        tab = getattr(self, '_ema_tab', None)
        if tab is None or tab['key'] != key:
            chunk = _lib.load().tdr_optim_chunk()
            ct, ci = [], []
            for t, (a, _) in enumerate(pairs):
                n = (a.numel() + chunk - 1) // chunk
                ct += [t] * n
                ci += list(range(n))
            dev = pairs[0][0].device
            tab = self._ema_tab = dict(
                key=key, n=len(ct),
                src=torch.tensor([k[0] for k in key], dtype=torch.int64).to(dev),
                dst=torch.tensor([k[1] for k in key], dtype=torch.int64).to(dev),
                sizes=torch.tensor([a.numel() for a, _ in pairs], dtype=torch.int64).to(dev),
                ct=torch.tensor(ct, dtype=torch.int32).to(dev), ci=torch.tensor(ci, dtype=torch.int32).to(dev))
        _lib.check(_lib.load().tdr_multi_ema(tab['src'].data_ptr(), tab['dst'].data_ptr(), tab['sizes'].data_ptr(),
                                             tab['ct'].data_ptr(), tab['ci'].data_ptr(), tab['n'], float(decay), K._stream()),
                   'tdr_multi_ema')

    def get_current_log(self):
        """floats, like the reference; the device->host read happens here (at print_freq), not in
        every optimize_parameters call."""
        out = OrderedDict((k, float(v)) for k, v in self.log_dict.items())
        if any(v != v or v in (float('inf'), float('-inf')) for v in out.values()):
            from .. import kernels as K
            hint = (' -- under TDR_MATH=hx2 the forward convolutions need activations inside the fp16 range (|x| < 65504); '
                    'TDR_MATH=bx3 has the full fp32 range') if K.fp16_path() else ''
            raise FloatingPointError(f'non-finite loss {dict(out)}{hint}')
        return out

    def model_to_device(self, net):
        """to(device); in distributed mode a GradAllReducer (RCCL over xGMI) takes DDP's place -- including what the
        DistributedDataParallel constructor does first (reference :76-82): every rank starts from rank 0's parameters
        and buffers.  The trainer seeds rank r with manual_seed + r (main_train_restoration_with_ref_input.py:55), so
        without the broadcast the replicas would start from different random weights."""
        net = net.to(self.device)
        if self.opt['dist']:
            self.sync_from_rank0(net)
            self.grad_reducer = GradAllReducer(list(net.named_parameters()),
                                               bucket_mb=self.opt.get('dist_bucket_mb', 64))
        return net

    def sync_from_rank0(self, net):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        from ..parallel import data_plane
        comm = data_plane()
        with torch.no_grad():
            for t in list(self.get_bare_model(net).parameters()) + list(self.get_bare_model(net).buffers()):
                if comm is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
                    comm.broadcast(t.data, root=0)
                else:
                    dist.broadcast(t.data, src=0)

    def setup_schedulers(self):
        train_opt = self.opt['train']
        scheduler_type = train_opt['scheduler'].pop('type')
        table = {
            'MultiStepLR': lr_scheduler.MultiStepRestartLR, 'MultiStepRestartLR': lr_scheduler.MultiStepRestartLR,
            'CosineAnnealingRestartLR': lr_scheduler.CosineAnnealingRestartLR,
            'CosineAnnealingRestartCyclicLR': lr_scheduler.CosineAnnealingRestartCyclicLR,
        }
        for optimizer in self.optimizers:
            if scheduler_type in table:
                self.schedulers.append(table[scheduler_type](optimizer, **train_opt['scheduler']))
            elif scheduler_type == 'TrueCosineAnnealingLR':
                self.schedulers.append(torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, **train_opt['scheduler']))
            elif scheduler_type == 'LinearLR':
                self.schedulers.append(lr_scheduler.LinearLR(optimizer, train_opt['total_iter']))
            else:
                raise NotImplementedError(f'Scheduler {scheduler_type} is not implemented yet.')

    def get_bare_model(self, net):
        return net.module if hasattr(net, 'module') and isinstance(net.module, torch.nn.Module) else net

    @master_only
    def print_network(self, net):
        net = self.get_bare_model(net)
        n = sum(p.numel() for p in net.parameters())
        logger.info(f'Network: {net.__class__.__name__}, with parameters: {n:,d}')

    def _set_lr(self, lr_groups_l):
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for param_group, lr in zip(optimizer.param_groups, lr_groups):
                param_group['lr'] = lr

    def _get_init_lr(self):
        return [[v['initial_lr'] for v in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        if current_iter > 1:
            for scheduler in self.schedulers:
                scheduler.step()
        if current_iter < warmup_iter:
            self._set_lr([[v / warmup_iter * current_iter for v in g] for g in self._get_init_lr()])

    def get_current_learning_rate(self):
        return [g['lr'] for g in self.optimizers[0].param_groups]

    @master_only
    def save_network(self, net, net_label, current_iter, param_key='params'):
        if current_iter == -1:
            current_iter = 'latest'
        save_path = os.path.join(self.opt['path']['models'], f'{net_label}_{current_iter}.pth')
        nets = net if isinstance(net, list) else [net]
        keys = param_key if isinstance(param_key, list) else [param_key]
        assert len(nets) == len(keys), 'The lengths of net and param_key should be the same.'
        save_dict = {}
        for n, k in zip(nets, keys):
            sd = OrderedDict()
            for name, p in self.get_bare_model(n).state_dict().items():
                sd[name[7:] if name.startswith('module.') else name] = p.cpu()
            save_dict[k] = sd
        torch.save(save_dict, save_path)

    def load_network(self, net, load_path, strict=True, param_key='params'):
        net = self.get_bare_model(net)
        logger.info(f'Loading {net.__class__.__name__} model from {load_path}.')
        load_net = torch.load(load_path, map_location='cpu')
        if param_key is not None:
            if param_key not in load_net and 'params' in load_net:
                param_key = 'params'
            load_net = load_net[param_key]
        for k in list(load_net.keys()):
            if k.startswith('module.'):
                load_net[k[7:]] = load_net.pop(k)
        if not strict:
            crt = net.state_dict()
            for k in set(crt) & set(load_net):
                if crt[k].size() != load_net[k].size():
                    logger.warning(f'Size different, ignore [{k}]')
                    load_net[k + '.ignore'] = load_net.pop(k)
        net.load_state_dict(load_net, strict=strict)
        if self.opt.get('dist'):
            self.sync_from_rank0(net)      # a non-strict partial load leaves the unmatched tensors rank-specific

    @master_only
    def save_training_state(self, epoch, current_iter):
        if current_iter != -1:
            state = {'epoch': epoch, 'iter': current_iter,
                     'optimizers': [o.state_dict() for o in self.optimizers],
                     'schedulers': [s.state_dict() for s in self.schedulers]}
            if hasattr(self, 'extra_training_state'):
                state['tdr'] = self.extra_training_state()
            torch.save(state, os.path.join(self.opt['path']['training_states'], f'{current_iter}.state'))

    def resume_training(self, resume_state):
        ro, rs = resume_state['optimizers'], resume_state['schedulers']
        assert len(ro) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(rs) == len(self.schedulers), 'Wrong lengths of schedulers'
        for i, o in enumerate(ro):
            self.optimizers[i].load_state_dict(o)
        for i, s in enumerate(rs):
            self.schedulers[i].load_state_dict(s)
        if hasattr(self, 'load_extra_training_state'):
            self.load_extra_training_state(resume_state.get('tdr'))

    def reduce_loss_dict(self, loss_dict):
        """dist.reduce to rank 0 then / world (reference :353-378); returns python floats."""
        with torch.no_grad():
            if self.opt['dist']:
                rank, world = get_dist_info()
                keys = list(loss_dict.keys())
                losses = torch.stack([loss_dict[k].reshape(()) for k in keys], 0)
                losses = reduce_loss_to_rank0(losses, world, rank)
                loss_dict = {k: v for k, v in zip(keys, losses)}
            return OrderedDict((k, v.reshape(())) for k, v in loss_dict.items())

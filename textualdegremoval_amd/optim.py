"""Global-norm clip + AdamW on the HIP multi-tensor kernels.

Replaces `torch.nn.utils.clip_grad_norm_(net_g.parameters(), 0.01)` followed by
`torch.optim.AdamW.step()` (reference models/image_restoration_ref_model.py:
172-178, 276-279).  It is a torch.optim.Optimizer subclass whose state_dict uses
AdamW's key names ('step', 'exp_avg', 'exp_avg_sq'), so the reference's
`.state` checkpoints (models/base_model.py:311-351) load unchanged.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check


class FusedClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.01,
                 use_grad_clip=True):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        super().__init__(params, defaults)
        if len(self.param_groups) > 4:
            raise NotImplementedError('at most 4 parameter groups')
        self.max_norm = float(max_norm)
        self.use_grad_clip = bool(use_grad_clip)
        self._tables = None
        self.last_sumsq = None
        self._step_count = None
        self._hp_host = None
        self._hp_dev = None

    # -- pointer / chunk tables (rebuilt when gradients or state tensors are re-allocated)
    def _build(self):
        lib = _lib.load()
        chunk = lib.tdr_optim_chunk()
        ps, group_of = [], []
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                if p.grad is None:
                    continue
                ps.append(p)
                group_of.append(gi)
        dev = ps[0].device
        for p in ps:
            st = self.state[p]
            if 'exp_avg' not in st:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            assert p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(),
                     self.state[p]['exp_avg_sq'].data_ptr()) for p in ps)
        if self._tables is not None and self._tables['key'] == key:
            return self._tables
        i64 = torch.int64
        sizes = torch.tensor([p.numel() for p in ps], dtype=i64)
        ct, ci = [], []
        for t, p in enumerate(ps):
            n = (p.numel() + chunk - 1) // chunk
            ct += [t] * n
            ci += list(range(n))
        tab = dict(
            key=key, ps=ps, n_chunks=len(ct),
            params=torch.tensor([k[0] for k in key], dtype=i64).to(dev),
            grads=torch.tensor([k[1] for k in key], dtype=i64).to(dev),
            m=torch.tensor([k[2] for k in key], dtype=i64).to(dev),
            v=torch.tensor([k[3] for k in key], dtype=i64).to(dev),
            sizes=sizes.to(dev), group=torch.tensor(group_of, dtype=torch.int32).to(dev),
            chunk_tensor=torch.tensor(ct, dtype=torch.int32).to(dev), chunk_index=torch.tensor(ci, dtype=torch.int32).to(dev),
            partial=torch.empty(len(ct), dtype=torch.float64, device=dev),
            sumsq=torch.zeros(1, dtype=torch.float64, device=dev),
        )
        self._tables = tab
        return tab

    # -- the step is split so the kernel launches can live in a captured hipGraph:
    #    prepare() refreshes the 24-byte device block {lr[0..3], 1-b1^t, sqrt(1-b2^t)} (host side, every step),
    #    launch() enqueues the two multi-tensor kernels with step-invariant arguments.
    def prepare(self):
        t = self._build()
        g0 = self.param_groups[0]
        for g in self.param_groups:
            assert tuple(g['betas']) == tuple(g0['betas']) and g['eps'] == g0['eps'] and \
                g['weight_decay'] == g0['weight_decay'], 'groups may differ in lr only'
        if self._step_count is None:
            self._step_count = int(self.state[t['ps'][0]]['step'])
        self._step_count += 1
        step = self._step_count
        b1, b2 = float(g0['betas'][0]), float(g0['betas'][1])
        lrs = [float(g['lr']) for g in self.param_groups] + [0.0] * (4 - len(self.param_groups))
        if self._hp_dev is None:
            self._hp_dev = torch.empty(6, dtype=torch.float32, device=t['ps'][0].device)
        # a fresh pinned block per step: torch's host allocator will not recycle it before the async copy ran
        hp = torch.tensor(lrs + [1.0 - b1 ** step, (1.0 - b2 ** step) ** 0.5], dtype=torch.float32).pin_memory()
        self._hp_dev.copy_(hp, non_blocking=True)

    def launch(self):
        lib = _lib.load()
        t = self._build()
        g0 = self.param_groups[0]
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.tdr_grad_sumsq(t['grads'].data_ptr(), t['sizes'].data_ptr(), t['chunk_tensor'].data_ptr(),
                                 t['chunk_index'].data_ptr(), t['n_chunks'], t['partial'].data_ptr(),
                                 t['sumsq'].data_ptr(), stream), 'tdr_grad_sumsq')
        check(lib.tdr_adamw_step_dev(t['params'].data_ptr(), t['grads'].data_ptr(), t['m'].data_ptr(), t['v'].data_ptr(),
                                     t['sizes'].data_ptr(), t['group'].data_ptr(), t['chunk_tensor'].data_ptr(),
                                     t['chunk_index'].data_ptr(), t['n_chunks'], t['sumsq'].data_ptr(),
                                     self._hp_dev.data_ptr(), self.max_norm, 1 if self.use_grad_clip else 0,
                                     float(g0['betas'][0]), float(g0['betas'][1]), float(g0['eps']),
                                     float(g0['weight_decay']), stream), 'tdr_adamw_step_dev')
        self.last_sumsq = t['sumsq']

    @torch.no_grad()
    def step(self, closure=None):
        self.prepare()
        self.launch()
        return None

    def state_dict(self):
        """AdamW-compatible: the shared step counter is materialised into every per-parameter state."""
        if self._step_count is not None:
            for st in self.state.values():
                if 'step' in st:
                    st['step'] = torch.tensor(float(self._step_count))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_count = None
        self._tables = None

    def grad_norm(self):
        """total gradient L2 norm of the last step (device sync)."""
        return float(self.last_sumsq.sqrt().item()) if self.last_sumsq is not None else None

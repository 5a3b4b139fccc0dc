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


class StepGuard:
    """Device-resident TdrStepGuard (include/tdr.h): loss scale of the fp16-split backward pass, finite-norm verdict,
    applied-step counter and bias corrections.  Host reads/writes are explicit and synchronising (set-up, checkpoints,
    logging) -- the step itself only touches it from kernels."""
    WORDS = C.sizeof(_lib.TdrStepGuard) // 4

    def __init__(self, device, growth_interval=1000):
        self.buf = torch.zeros(self.WORDS, dtype=torch.int32, device=device)
        self.growth_interval = int(growth_interval)
        self.write(scale=1.0, max_scale=1.0, step=0)

    def data_ptr(self):
        return self.buf.data_ptr()

    def read(self):
        raw = self.buf.cpu().numpy().tobytes()
        return _lib.TdrStepGuard.from_buffer_copy(raw)

    def write(self, scale=None, max_scale=None, step=None):
        g = self.read()
        if max_scale is not None:
            g.max_scale = float(max_scale)
        if scale is not None:
            g.scale, g.inv_scale, g.good = float(scale), 1.0 / float(scale), 0
        if step is not None:
            g.step = int(step)
        g.growth_interval = self.growth_interval
        g.finite = 1
        import numpy as np
        self.buf.copy_(torch.from_numpy(np.frombuffer(bytes(g), dtype=np.int32).copy()))

    def set_never_skip(self, on):
        """on: no finite-norm verdict -- every step is applied, as the reference's clip_grad_norm_ + AdamW.step() (the modes without
        a loss scale); the struct is then only the device-resident step counter.  Host write only when the setting changes."""
        if self.growth_interval > 0:
            self._growth_on = self.growth_interval
        want = -1 if on else getattr(self, '_growth_on', 1000)
        if want != self.growth_interval:
            self.growth_interval = want
            self.write()

    @property
    def never_skip(self):
        return self.growth_interval < 0

    def set_max_scale(self, s):
        """(re)start from the scale `s` when the caller's upper bound changes (first step, new input shape)"""
        if self.read().max_scale != float(s):
            self.write(scale=s, max_scale=s)


class FusedClipAdamW(torch.optim.Optimizer):
    """coupled_decay=True turns the update into torch.optim.Adam's (weight_decay * p added to the gradient), the
    reference's `optim_g.type: Adam` branch (image_restoration_ref_model.py:176-178)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.01,
                 use_grad_clip=True, coupled_decay=False):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        super().__init__(params, defaults)
        if len(self.param_groups) > 4:
            raise NotImplementedError('at most 4 parameter groups')
        self.max_norm = float(max_norm)
        self.use_grad_clip = bool(use_grad_clip)
        self.coupled_decay = bool(coupled_decay)
        self.frozen_groups = set()       # indices of param groups whose tensors are neither clipped nor updated
        self._tables = None
        self.last_sumsq = None
        self._hp_dev = None
        self.guard = None
        self._pending_step = None

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
        group_of = [-1 if gi in self.frozen_groups else gi for gi in group_of]
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(),
                     self.state[p]['exp_avg_sq'].data_ptr(), gi) for p, gi in zip(ps, group_of))
        if self._tables is not None and self._tables['key'] == key:
            return self._tables
        if self.guard is None:
            self.guard = StepGuard(dev)
        if self._pending_step is None:
            self._pending_step = int(self.state[ps[0]]['step'])
        if self._pending_step is not None and self._pending_step >= 0:
            self.guard.write(step=self._pending_step)
            self._pending_step = -1
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

    def set_frozen_groups(self, groups):
        groups = set(groups)
        if groups != self.frozen_groups:
            self.frozen_groups = groups
            self._tables = None

    # -- the step is split so the kernel launches can live in a captured hipGraph:
    #    prepare() refreshes the 16-byte device block {lr[0..3]} (host side, every step: the schedulers move the LRs),
    #    launch() enqueues the multi-tensor kernels with step-invariant arguments; the step count, the bias corrections
    #    and the finite-norm verdict live in the device-resident StepGuard.
    def prepare(self):
        t = self._build()
        g0 = self.param_groups[0]
        for g in self.param_groups:
            assert tuple(g['betas']) == tuple(g0['betas']) and g['eps'] == g0['eps'] and \
                g['weight_decay'] == g0['weight_decay'], 'groups may differ in lr only'
        lrs = [float(g['lr']) for g in self.param_groups] + [0.0] * (4 - len(self.param_groups))
        if self._hp_dev is None:
            self._hp_dev = torch.empty(4, dtype=torch.float32, device=t['ps'][0].device)
        # a fresh pinned block per step: torch's host allocator will not recycle it before the async copy ran
        hp = torch.tensor(lrs, dtype=torch.float32).pin_memory()
        self._hp_dev.copy_(hp, non_blocking=True)

    def launch(self):
        lib = _lib.load()
        t = self._build()
        g0 = self.param_groups[0]
        stream = torch.cuda.current_stream().cuda_stream
        b1, b2 = float(g0['betas'][0]), float(g0['betas'][1])
        check(lib.tdr_grad_sumsq_guarded(t['grads'].data_ptr(), t['sizes'].data_ptr(), t['group'].data_ptr(),
                                         t['chunk_tensor'].data_ptr(), t['chunk_index'].data_ptr(), t['n_chunks'],
                                         t['partial'].data_ptr(), t['sumsq'].data_ptr(), self.guard.data_ptr(), b1, b2, stream),
              'tdr_grad_sumsq_guarded')
        check(lib.tdr_adamw_step_guarded(t['params'].data_ptr(), t['grads'].data_ptr(), t['m'].data_ptr(), t['v'].data_ptr(),
                                         t['sizes'].data_ptr(), t['group'].data_ptr(), t['chunk_tensor'].data_ptr(),
                                         t['chunk_index'].data_ptr(), t['n_chunks'], t['sumsq'].data_ptr(),
                                         self._hp_dev.data_ptr(), self.guard.data_ptr(), self.max_norm,
                                         1 if self.use_grad_clip else 0, 1 if self.coupled_decay else 0, b1, b2,
                                         float(g0['eps']), float(g0['weight_decay']), stream), 'tdr_adamw_step_guarded')
        self.last_sumsq = t['sumsq']

    @torch.no_grad()
    def step(self, closure=None):
        self.prepare()
        self.launch()
        return None

    def ensure_guard(self, device):
        if self.guard is None:
            self.guard = StepGuard(device)
        return self.guard

    def applied_steps(self):
        """AdamW's t: steps actually applied (skipped non-finite steps do not count).  Device sync."""
        return int(self.guard.read().step) if self.guard is not None else 0

    def skipped_steps(self):
        return int(self.guard.read().skipped) if self.guard is not None else 0

    def state_dict(self):
        """AdamW-compatible: the shared step counter is materialised into every per-parameter state."""
        if self.guard is not None and self._pending_step == -1:
            n = self.applied_steps()
            for st in self.state.values():
                if 'step' in st:
                    st['step'] = torch.tensor(float(n))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._pending_step = None        # re-read from the loaded per-parameter 'step' at the next _build()
        self._tables = None

    def grad_norm(self):
        """total gradient L2 norm of the last step (device sync)."""
        return float(self.last_sumsq.sqrt().item()) if self.last_sumsq is not None else None

"""LR schedulers with the reference's names and closed forms (models/lr_scheduler.py).
Host-side scalar math; CosineAnnealingRestartCyclicLR is the one every YAML selects."""
import math
from collections import Counter

from torch.optim.lr_scheduler import _LRScheduler


def get_position_from_periods(iteration, cumulative_period):
    for i, period in enumerate(cumulative_period):
        if iteration <= period:
            return i


class _Restart(_LRScheduler):
    def __init__(self, optimizer, periods, restart_weights=(1,), last_epoch=-1):
        self.periods, self.restart_weights = periods, restart_weights
        assert len(periods) == len(restart_weights), 'periods and restart_weights should have the same length.'
        self.cumulative_period = [sum(periods[:i + 1]) for i in range(len(periods))]
        super().__init__(optimizer, last_epoch)

    def _cycle(self):
        idx = get_position_from_periods(self.last_epoch, self.cumulative_period)
        start = 0 if idx == 0 else self.cumulative_period[idx - 1]
        return idx, (self.last_epoch - start) / self.periods[idx]


class CosineAnnealingRestartLR(_Restart):
    def __init__(self, optimizer, periods, restart_weights=(1,), eta_min=0, last_epoch=-1):
        self.eta_min = eta_min
        super().__init__(optimizer, periods, restart_weights, last_epoch)

    def get_lr(self):
        idx, frac = self._cycle()
        w = self.restart_weights[idx]
        return [self.eta_min + w * 0.5 * (b - self.eta_min) * (1 + math.cos(math.pi * frac)) for b in self.base_lrs]


class CosineAnnealingRestartCyclicLR(_Restart):
    def __init__(self, optimizer, periods, restart_weights=(1,), eta_mins=(0,), last_epoch=-1):
        self.eta_mins = eta_mins
        super().__init__(optimizer, periods, restart_weights, last_epoch)

    def get_lr(self):
        idx, frac = self._cycle()
        w, eta = self.restart_weights[idx], self.eta_mins[idx]
        return [eta + w * 0.5 * (b - eta) * (1 + math.cos(math.pi * frac)) for b in self.base_lrs]


class MultiStepRestartLR(_LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, restarts=(0,), restart_weights=(1,), last_epoch=-1):
        self.milestones, self.gamma = Counter(milestones), gamma
        self.restarts, self.restart_weights = restarts, restart_weights
        assert len(restarts) == len(restart_weights), 'restarts and their weights do not match.'
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch in self.restarts:
            w = self.restart_weights[self.restarts.index(self.last_epoch)]
            return [g['initial_lr'] * w for g in self.optimizer.param_groups]
        if self.last_epoch not in self.milestones:
            return [g['lr'] for g in self.optimizer.param_groups]
        return [g['lr'] * self.gamma ** self.milestones[self.last_epoch] for g in self.optimizer.param_groups]


class LinearLR(_LRScheduler):
    def __init__(self, optimizer, total_iter, last_epoch=-1):
        self.total_iter = total_iter
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        w = 1 - self.last_epoch / self.total_iter
        return [w * g['initial_lr'] for g in self.optimizer.param_groups]

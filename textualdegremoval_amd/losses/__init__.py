"""Pixel losses with the reference's names (losses/losses.py:26-122).  On the device every criterion with mean reduction is
the HIP kernel tdr_pixel_loss (value + gradient in one pass); `step_kind()` tells the train step which one to fuse.  The
torch expressions below serve host tensors and the weighted / 'none' / 'sum' variants no shipped YAML selects."""
import numpy as np
import torch
from torch import nn

from .. import kernels as K

_reduction_modes = ['none', 'mean', 'sum']


class _PixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind, loss_weight, eps):
        loss, dpred = K.pixel_loss(kind, pred.contiguous(), target.contiguous(), loss_weight, eps)
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None, None


def _weight_reduce(loss, weight, reduction):
    """losses/loss_util.py:25-54 (weight_reduce_loss): element-wise weight, then the reduction; the weighted mean divides by
    the weight mass -- weight.sum() for a per-channel weight, weight.sum() * C for a one-channel weight (no clamp)."""
    if weight is not None:
        assert weight.dim() == loss.dim()
        assert weight.size(1) == 1 or weight.size(1) == loss.size(1)
        loss = loss * weight
    if weight is None or reduction == 'sum':
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    if reduction == 'mean':
        return loss.sum() / (weight.sum() if weight.size(1) > 1 else weight.sum() * loss.size(1))
    return loss


def _on_device(pred, target):
    return pred.is_cuda and pred.dim() == 4 and pred.dtype == torch.float32 and target.dtype == torch.float32


class L1Loss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction='mean'):
        super().__init__()
        if reduction not in _reduction_modes:
            raise ValueError(f'Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}')
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred, target, weight=None, **kwargs):
        if _on_device(pred, target) and weight is None and self.reduction == 'mean':
            return _PixFn.apply(pred, target, K.LOSS_L1, float(self.loss_weight), 0.0)
        return self.loss_weight * _weight_reduce((pred - target).abs(), weight, self.reduction)

    def step_kind(self):
        return (K.LOSS_L1, float(self.loss_weight), 0.0) if self.reduction == 'mean' else None


class MSELoss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction='mean'):
        super().__init__()
        if reduction not in _reduction_modes:
            raise ValueError(f'Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}')
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred, target, weight=None, **kwargs):
        if _on_device(pred, target) and weight is None and self.reduction == 'mean':
            return _PixFn.apply(pred, target, K.LOSS_MSE, float(self.loss_weight), 0.0)
        return self.loss_weight * _weight_reduce((pred - target) ** 2, weight, self.reduction)

    def step_kind(self):
        return (K.LOSS_MSE, float(self.loss_weight), 0.0) if self.reduction == 'mean' else None


class PSNRLoss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction='mean', toY=False):
        super().__init__()
        assert reduction == 'mean'
        self.loss_weight, self.scale, self.toY = loss_weight, 10 / np.log(10), toY

    def step_kind(self):
        return (K.LOSS_PSNR_Y if self.toY else K.LOSS_PSNR, float(self.loss_weight), 0.0)

    def forward(self, pred, target):
        assert len(pred.size()) == 4
        if _on_device(pred, target) and (not self.toY or pred.shape[1] == 3):
            return _PixFn.apply(pred, target, *self.step_kind())
        if self.toY:
            coef = torch.tensor([65.481, 128.553, 24.966], device=pred.device).reshape(1, 3, 1, 1)
            pred = ((pred * coef).sum(dim=1, keepdim=True) + 16.) / 255.
            target = ((target * coef).sum(dim=1, keepdim=True) + 16.) / 255.
        return self.loss_weight * self.scale * torch.log(((pred - target) ** 2).mean(dim=(1, 2, 3)) + 1e-8).mean()


class CharbonnierLoss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction='mean', eps=1e-3):
        super().__init__()
        self.eps = eps
        self.loss_weight, self.reduction = loss_weight, reduction      # stored for the step; the value ignores both (reference :114-122)

    def step_kind(self):
        return (K.LOSS_CHARBONNIER, 1.0, float(self.eps))

    def forward(self, x, y):
        if _on_device(x, y):
            return _PixFn.apply(x, y, *self.step_kind())
        diff = x - y
        return torch.mean(torch.sqrt(diff * diff + self.eps * self.eps))

"""CPU restatement (numpy float64) of the reference's pixel criteria, value and d(loss)/d(pred) -- TEST INFRASTRUCTURE
ONLY (imported by tests/; the product computes these in csrc/tdr_pointwise.hip tdr_pixel_loss and has no host fallback on
the train step).  Pinned by tests/golden/losses.npz, produced by running the reference's losses/losses.py itself
(tests/golden/make_golden_losses.py)."""
import numpy as np

SCALE = 10.0 / np.log(10.0)
COEF = np.array([65.481, 128.553, 24.966]).reshape(1, 3, 1, 1)


def l1(pred, target, loss_weight=1.0):
    """L1Loss, reduction='mean' (losses/losses.py:26-53)"""
    d = pred.astype(np.float64) - target
    return loss_weight * np.abs(d).mean(), loss_weight * np.sign(d) / d.size


def mse(pred, target, loss_weight=1.0):
    """MSELoss, reduction='mean' (:55-82)"""
    d = pred.astype(np.float64) - target
    return loss_weight * (d * d).mean(), loss_weight * 2.0 * d / d.size


def charbonnier(pred, target, eps=1e-3):
    """CharbonnierLoss (:111-122): mean sqrt(d^2 + eps^2); loss_weight is not applied there"""
    d = pred.astype(np.float64) - target
    r = np.sqrt(d * d + eps * eps)
    return r.mean(), d / r / d.size


def psnr(pred, target, loss_weight=1.0, toY=False):
    """PSNRLoss (:84-109): w * 10/ln10 * mean_n log(mean_chw d^2 + 1e-8); toY: BT.601 luma / 255 first"""
    p, t = pred.astype(np.float64), target.astype(np.float64)
    if toY:
        p = ((p * COEF).sum(axis=1, keepdims=True) + 16.) / 255.
        t = ((t * COEF).sum(axis=1, keepdims=True) + 16.) / 255.
    d = p - t
    n = d.shape[0]
    m = (d * d).reshape(n, -1).mean(axis=1)
    loss = loss_weight * SCALE * np.log(m + 1e-8).mean()
    g = loss_weight * SCALE / n / (m + 1e-8).reshape(n, 1, 1, 1) * 2.0 * d / d[0].size
    if toY:
        g = g * COEF / 255.
    return loss, g


KINDS = {'l1': l1, 'mse': mse, 'charbonnier': charbonnier, 'psnr': psnr, 'psnr_y': lambda p, t, w=1.0: psnr(p, t, w, toY=True)}

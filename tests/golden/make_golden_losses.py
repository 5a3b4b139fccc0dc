"""Generate the pixel-criterion golden vectors by running the REFERENCE's losses/losses.py on CPU.

Run in the build container only (needs /root/reference, which never travels):
    python tests/golden/make_golden_losses.py
Writes tests/golden/losses.npz: seeded pred / target, and for every criterion the reference's loss value and its autograd
gradient with respect to pred.  Nothing from the reference is copied: it is imported, executed, only numbers are saved."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
from losses.losses import CharbonnierLoss, L1Loss, MSELoss, PSNRLoss  # noqa: E402

rng = np.random.default_rng(20260928)
target = rng.random((3, 3, 20, 28)).astype(np.float32)
pred = (target + rng.normal(0, 0.08, target.shape)).astype(np.float32)
pred[1] = target[1] + rng.normal(0, 0.004, target[1].shape).astype(np.float32)       # one nearly clean image (PSNRLoss weights it up)
out = {'pred': pred, 'target': target}
CASES = {'l1': L1Loss(loss_weight=0.7), 'mse': MSELoss(loss_weight=1.3), 'charbonnier': CharbonnierLoss(loss_weight=5.0, eps=1e-3),
         'charbonnier_eps2': CharbonnierLoss(eps=0.05), 'psnr': PSNRLoss(loss_weight=0.5), 'psnr_y': PSNRLoss(loss_weight=1.0, toY=True)}
for name, crit in CASES.items():
    p = torch.tensor(pred, requires_grad=True)
    loss = crit(p, torch.tensor(target))
    loss.backward()
    out[name + '_loss'] = np.float64(loss.item())
    out[name + '_grad'] = p.grad.numpy()
# weighted / non-mean reductions (losses/loss_util.py:25-54 weight_reduce_loss): one-channel and per-channel weights
w1 = rng.random((3, 1, 20, 28)).astype(np.float32)
w3 = rng.random((3, 3, 20, 28)).astype(np.float32)
out['w1'], out['w3'] = w1, w3
for name, crit, w in (('l1_w1_mean', L1Loss(loss_weight=0.7), w1), ('l1_w3_mean', L1Loss(), w3), ('mse_w1_mean', MSELoss(loss_weight=2.0), w1),
                      ('mse_w3_sum', MSELoss(reduction='sum'), w3), ('l1_sum', L1Loss(reduction='sum'), None),
                      ('l1_w1_none', L1Loss(reduction='none'), w1)):
    p = torch.tensor(pred, requires_grad=True)
    loss = crit(p, torch.tensor(target), weight=None if w is None else torch.tensor(w))
    out[name + '_loss'] = loss.detach().numpy().astype(np.float64)
    loss.sum().backward()
    out[name + '_grad'] = p.grad.numpy()
np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)
print({k: float(np.sum(v)) for k, v in out.items() if k.endswith('_loss')})

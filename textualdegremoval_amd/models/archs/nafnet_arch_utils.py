"""LayerNorm2d on the HIP kernels.  Mirrors models/archs/nafnet_arch_utils.py:264-300
of the reference (same class names, parameters `weight`/`bias`, eps 1e-6)."""
import torch
import torch.nn as nn

from ... import kernels as K


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: the HIP path needs tensors on the MI355X; there is no CPU fallback '
                           '(the CPU oracle lives in oracle/ and is test infrastructure only).')


class LayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        require_gpu(x, 'LayerNorm2d')
        x = x.contiguous()
        y, mu, rstd = K.layernorm2d_fwd(x, weight, bias, eps)
        ctx.save_for_backward(x, mu, rstd, weight)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        x, mu, rstd, weight = ctx.saved_tensors
        gx, gw, gb = K.layernorm2d_bwd(grad_output.contiguous(), x, mu, rstd, weight)
        return gx, gw, gb, None


class LayerNorm2d(nn.Module):
    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.register_parameter('weight', nn.Parameter(torch.ones(channels)))
        self.register_parameter('bias', nn.Parameter(torch.zeros(channels)))
        self.eps = eps

    def forward(self, x):
        return LayerNormFunction.apply(x, self.weight, self.bias, self.eps)

"""TLSC (test-time local statistics converter) pieces -- the operator surface of models/archs/nafnet_local_arch.py of the reference:
`AvgPool2d` (the local box mean that replaces nn.AdaptiveAvgPool2d(1) at inference, :10-75) on the HIP kernel of
csrc/tdr_tlsc.hip, `replace_layers` and `Local_Base` (:77-104).  The guided / un-guided NAFNet classes of this package do not
carry nn.AdaptiveAvgPool2d children (their blocks run as fused kernels), so `NAFNetLocal` fixes its kernels arithmetically
(engine.tlsc_kernel_sizes); these classes serve module code written against the reference's names."""
import torch
import torch.nn as nn

from ... import kernels as K
from .nafnet_arch_utils import require_gpu


class AvgPool2d(nn.Module):
    def __init__(self, kernel_size=None, base_size=None, auto_pad=True, fast_imp=False, train_size=None):
        super().__init__()
        self.kernel_size, self.base_size, self.auto_pad = kernel_size, base_size, auto_pad
        self.fast_imp, self.train_size = fast_imp, train_size
        if fast_imp:
            raise NotImplementedError('AvgPool2d(fast_imp=True): the sub-sampled variant (:46-60) is not built')
        if not auto_pad:
            raise NotImplementedError('AvgPool2d(auto_pad=False): the kernel writes the replicate-padded map (every reference use pads)')

    def extra_repr(self):
        return 'kernel_size={}, base_size={}, stride={}, fast_imp={}'.format(self.kernel_size, self.base_size, self.kernel_size, self.fast_imp)

    def forward(self, x):
        require_gpu(x, 'AvgPool2d')
        if self.kernel_size is None and self.base_size:                      # fixed by the first forward (:29-36)
            if isinstance(self.base_size, int):
                self.base_size = (self.base_size, self.base_size)
            ts = self.train_size
            self.kernel_size = [x.shape[2] * self.base_size[0] // ts[-2], x.shape[3] * self.base_size[1] // ts[-1]]
        x = x.detach().contiguous()
        N, Cc, H, W = x.shape
        if self.kernel_size[0] >= H and self.kernel_size[1] >= W:            # F.adaptive_avg_pool2d(x, 1) (:43-44)
            return K.plane_mean(x).view(N, Cc, 1, 1)
        return K.local_avgpool(x, int(self.kernel_size[0]), int(self.kernel_size[1]))


def replace_layers(model, base_size, train_size, fast_imp, **kwargs):
    for n, m in model.named_children():
        if len(list(m.children())) > 0:
            replace_layers(m, base_size, train_size, fast_imp, **kwargs)
        if isinstance(m, nn.AdaptiveAvgPool2d):
            assert m.output_size == 1
            setattr(model, n, AvgPool2d(base_size=base_size, fast_imp=fast_imp, train_size=train_size))


class Local_Base:
    def convert(self, *args, train_size, **kwargs):
        replace_layers(self, *args, train_size=train_size, **kwargs)
        imgs = torch.rand(train_size, device='cuda')
        with torch.no_grad():
            self.forward(imgs)

"""SFNet-ref (models/archs/network_sfnet_guided_arch.py:410-797 of the reference) -- registered so that the shipped YAML's
`type: SFNetRefFusion` resolves, but there is no network behind it, because there is none in the reference either (defect R8,
recorded by tests/golden/make_golden_defects.py by running the reference): the class constructs, and its first forward pass
raises for every width -- the MASA Encoder (:292-317) feeds nf-channel stride-2 convolutions into 2nf / 4nf-channel residual
blocks, returns three feature levels where forward reads feat[4] (:621), and EBlockResFusion.forward (:180-186) never calls
its layers.  With no runnable reference there is no oracle, no golden vector and nothing to be in parity with; the same
RuntimeError is raised here at the same point (the first forward)."""
from torch import nn


class SFNetRefFusion(nn.Module):
    def __init__(self, mode, num_res=16, nf=64, ext_n_blocks=(4, 4, 4, 4), reffusion_n_blocks=(1, 1, 1, 1),
                 reffusion_n_blocks_middle=1, scale=1, num_nbr=1, psize=3, lr_block_size=8, ref_down_block_size=1.5,
                 dilations=(1, 2, 3)):
        super().__init__()
        self.cfg = dict(mode=mode, num_res=num_res, nf=nf, ext_n_blocks=list(ext_n_blocks),
                        reffusion_n_blocks=list(reffusion_n_blocks), reffusion_n_blocks_middle=reffusion_n_blocks_middle,
                        scale=scale, num_nbr=num_nbr, psize=psize, lr_block_size=lr_block_size,
                        ref_down_block_size=ref_down_block_size, dilations=list(dilations))

    def forward(self, x, ref):
        nf = self.cfg['nf']
        raise RuntimeError(f'Given groups=1, weight of size [{2 * nf}, {2 * nf}, 3, 3], expected input to have {2 * nf} channels, '
                           f'but got {nf} channels instead (SFNetRefFusion cannot run a forward pass in the reference: defect R8)')

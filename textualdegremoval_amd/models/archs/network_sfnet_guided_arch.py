"""The SFNet file of the reference (models/archs/network_sfnet_guided_arch.py).

`SFNet` (:320-407) -- the un-guided network -- is built here: same constructor (`mode`, `num_res`), same module tree, registration
order, parameter / buffer names and default initialisation (the layers below are ordinary torch modules used as PARAMETER CONTAINERS; no
torch operator runs in the forward or backward pass: both go through textualdegremoval_amd/sfnet_engine.py on the HIP kernels).  Only
mode[0] == 'train' (global average pools) runs in training mode (BatchNorm2d on batch statistics, running buffers moved) and after .eval()
(BatchNorm2d on its running statistics: the validation pass of the trainer); mode[0] == 'test' -- the reference's inference network, where Gap,
Patch_ap and SFconv pool with the TLSC box mean of the Indoor / Outdoor base size (sfnet_arch_utils.py:108-113, :226-229, :247-250) -- runs
after .eval(), forward only.  (The reference caches each pooling window from the first input a module instance sees; here the window follows
the input of the call, which is the same thing for a fresh network or inputs of one size.)

`SFNetRefFusion` (:410-797) is registered so that the shipped YAML's `type: SFNetRefFusion` resolves, but there is no network behind it,
because there is none in the reference either (defect R8, recorded by tests/golden/make_golden_defects.py by running the reference): the
class constructs, and its first forward pass raises for every width -- the MASA Encoder (:292-317) feeds nf-channel stride-2
convolutions into 2nf / 4nf-channel residual blocks, returns three feature levels where forward reads feat[4] (:621), and
EBlockResFusion.forward (:180-186) never calls its layers.  The same RuntimeError is raised here at the same point."""
import torch
from torch import nn

BASE = 32


def _basic(cin, cout, k, stride=1, transpose=False):
    """BasicConv's container (sfnet_arch_utils.py:76-98): `main.0` is the convolution (bias on, no norm); the GELU has no parameters"""
    m = nn.Module()
    conv = nn.ConvTranspose2d(cin, cout, k, padding=k // 2 - 1, stride=stride, bias=True) if transpose else \
        nn.Conv2d(cin, cout, k, padding=k // 2, stride=stride, bias=True)
    m.main = nn.Sequential(conv)
    return m


def _dyn(c, k, group=8):
    """dynamic_filter's container (:152-174) with SFconv (:195-221) as `modulate`"""
    m = nn.Module()
    m.lamb_l = nn.Parameter(torch.zeros(c))                 # (registered by the reference, never used in its forward)
    m.lamb_h = nn.Parameter(torch.zeros(c))
    m.conv = nn.Conv2d(c, group * k * k, 1, bias=False)
    m.bn = nn.BatchNorm2d(group * k * k)
    nn.init.kaiming_normal_(m.conv.weight, mode='fan_out', nonlinearity='relu')
    d = max(c // 2, 32)
    mod = nn.Module()
    mod.fc = nn.Conv2d(c, d, 1)
    mod.fcs = nn.ModuleList([nn.Conv2d(d, c, 1), nn.Conv2d(d, c, 1)])
    mod.out = nn.Conv2d(c, c, 1)
    m.modulate = mod
    return m


def _res(c, filt):
    """ResBlock's container (:120-133)"""
    m = nn.Module()
    m.conv1 = _basic(c, c, 3)
    m.conv2 = _basic(c, c, 3)
    m.dyna = _dyn(c // 2, 3) if filt else nn.Identity()
    m.dyna_2 = _dyn(c // 2, 5) if filt else nn.Identity()
    la = nn.Module()
    la.h = nn.Parameter(torch.zeros(c // 2 * 4))
    la.l = nn.Parameter(torch.zeros(c // 2 * 4))
    m.localap = la
    ga = nn.Module()
    ga.fscale_d = nn.Parameter(torch.zeros(c // 2))
    ga.fscale_h = nn.Parameter(torch.zeros(c // 2))
    m.global_ap = ga
    return m


def _block(c, num_res):
    m = nn.Module()
    m.layers = nn.Sequential(*[_res(c, r == num_res - 1) for r in range(num_res)])
    return m


def _scm(c):
    m = nn.Module()
    m.main = nn.Sequential(_basic(3, c // 4, 3), _basic(c // 4, c // 2, 1), _basic(c // 2, c // 2, 3), _basic(c // 2, c, 1),
                           nn.InstanceNorm2d(c, affine=True))
    return m


def _fam(c):
    m = nn.Module()
    m.merge = _basic(2 * c, c, 3)
    return m


class _SFNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, num_res, names, buffers, *params):
        from ... import sfnet_engine as SE
        P = dict(zip(names, (p.detach() for p in params)))
        P.update(buffers)
        outs, saved = SE.net_fwd(P, x.detach(), num_res)
        ctx.saved_state, ctx.P, ctx.names = saved, P, names
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        from ... import sfnet_engine as SE
        G = SE.net_bwd([d.contiguous() for d in douts], ctx.P, ctx.saved_state)
        return (None, None, None, None) + tuple(G[n].reshape(ctx.P[n].shape) if n in G else None for n in ctx.names)


class SFNet(nn.Module):
    def __init__(self, mode, num_res=16):
        super().__init__()
        if mode[0] not in ('train', 'test') or (mode[0] == 'test' and mode[1] not in ('Indoor', 'Outdoor')):
            raise ValueError(f"SFNet: mode {mode!r} (['train', ...] or ['test', 'Indoor' | 'Outdoor'], sfnet_arch_utils.py:108-113)")
        # mode[0] == 'test': Gap / Patch_ap / SFconv pool with the TLSC box mean (base size 246 Indoor / 210 Outdoor); same parameters
        self.tlsc = {'Indoor': 246, 'Outdoor': 210}[mode[1]] if mode[0] == 'test' else None
        b = BASE
        self.num_res = num_res
        self.Encoder = nn.ModuleList([_block(b, num_res), _block(2 * b, num_res), _block(4 * b, num_res)])
        self.feat_extract = nn.ModuleList([_basic(3, b, 3), _basic(b, 2 * b, 3, 2), _basic(2 * b, 4 * b, 3, 2),
                                           _basic(4 * b, 2 * b, 4, 2, transpose=True), _basic(2 * b, b, 4, 2, transpose=True),
                                           _basic(b, 3, 3)])
        self.Decoder = nn.ModuleList([_block(4 * b, num_res), _block(2 * b, num_res), _block(b, num_res)])
        self.Convs = nn.ModuleList([_basic(4 * b, 2 * b, 1), _basic(2 * b, b, 1)])
        self.ConvsOut = nn.ModuleList([_basic(4 * b, 3, 3), _basic(2 * b, 3, 3)])
        self.FAM1 = _fam(4 * b)
        self.SCM1 = _scm(4 * b)
        self.FAM2 = _fam(2 * b)
        self.SCM2 = _scm(2 * b)

    def forward(self, x):
        """-> [out at 1/4, out at 1/2, out at full size] (reference :366-407); H and W multiples of 8 (two stride-2 levels, quadrants)"""
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError('SFNet: image height and width must be multiples of 8')
        named = list(self.named_parameters())
        buffers = {k: v for k, v in self.named_buffers()}
        if not self.training:
            # module.eval() (the trainer's validation pass): BatchNorm2d on its running statistics, no buffer moves; forward only --
            # the outputs carry no autograd graph (validation runs under torch.no_grad() in the reference, base_model / image_restoration_model)
            from ... import sfnet_engine as SE
            P = {k: p.detach() for k, p in named}
            P.update(buffers)
            return list(SE.net_fwd(P, x.detach(), self.num_res, training=False, tlsc=self.tlsc)[0])
        if self.tlsc is not None:
            raise NotImplementedError("SFNet mode 'test' is the reference's inference network: call .eval() first (no training pass is built for it)")
        return list(_SFNetFn.apply(x, self.num_res, [k for k, _ in named], buffers, *[p for _, p in named]))


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

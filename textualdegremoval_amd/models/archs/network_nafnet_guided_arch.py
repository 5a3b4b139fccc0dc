"""NAFNet guided architecture on the HIP engine.

Drop-in mirror of the reference's models/archs/network_nafnet_guided_arch.py:
same class names, constructor kwargs, forward signatures, parameter names,
registration order (= state-dict order, optimizer group order) and default
initialisation (the nn.Conv2d members are constructed exactly like the
reference's, so the RNG stream and init match; they are used as parameter
containers only -- their ATen forward is never called).  All arithmetic runs in
libtdr_hip.so through textualdegremoval_amd.engine.
"""
import functools

import torch
import torch.nn as nn

from ... import engine as E
from .nafnet_arch_utils import LayerNorm2d, require_gpu


def _named(module):
    names, params = [], []
    for k, p in module.named_parameters():
        names.append(k)
        params.append(p)
    return names, params


class _BlockFn(torch.autograd.Function):
    """one NAFBlock forward/backward (reference :216-238)."""

    @staticmethod
    def forward(ctx, x, names, c_out, *params):
        require_gpu(x, 'NAFBlock')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = E.naf_fwd(x.contiguous(), P, c_out)
        ctx.names, ctx.P, ctx.saved = names, P, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, G = E.naf_bwd(dout.contiguous(), ctx.P, ctx.saved)
        return (dx, None, None) + tuple(G[k] for k in ctx.names)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, names, n_blks, *params):
        require_gpu(x, 'Encoder')
        P = dict(zip(names, [p.detach() for p in params]))
        feats, saved = E.encoder_fwd(x.contiguous(), P, '', n_blks)
        ctx.names, ctx.P, ctx.saved, ctx.n_blks = names, P, saved, n_blks
        return tuple(feats)

    @staticmethod
    def backward(ctx, *dfeats):
        G = {}
        E.encoder_bwd([d.contiguous() if d is not None else None for d in dfeats], ctx.P, '', ctx.n_blks, ctx.saved, G)
        return (None, None, None) + tuple(G[k] for k in ctx.names)


class _NetFn(torch.autograd.Function):
    """whole NAFNetRefFusion forward/backward in one node: every parameter is used
    exactly once, so autograd performs no device arithmetic of its own."""

    @staticmethod
    def forward(ctx, inp, ref, names, cfg, *params):
        require_gpu(inp, 'NAFNetRefFusion')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = E.net_fwd(P, cfg, inp, ref)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = E.net_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[k] for k in ctx.names)


class _UNetFn(torch.autograd.Function):
    """the un-guided NAFNet as one autograd node (gradient w.r.t. the input image included)"""

    @staticmethod
    def forward(ctx, inp, names, cfg, *params):
        require_gpu(inp, 'NAFNet')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = E.unet_fwd(P, cfg, inp)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        dinp, G = E.unet_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (dinp, None, None) + tuple(G[k] for k in ctx.names)


def make_layer(block, n_layers):
    return nn.Sequential(*[block() for _ in range(n_layers)])


class ResidualBlock(nn.Module):
    def __init__(self, nf, kernel_size=3, stride=1, padding=1, dilation=1, act='relu'):
        super().__init__()
        if kernel_size != 3 or stride != 1 or padding != 1 or dilation != 1 or act != 'relu':
            raise NotImplementedError('HIP path: ResidualBlock is 3x3/s1/p1/ReLU (the only form the reference uses)')
        self.conv1 = nn.Conv2d(nf, nf, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation)
        self.conv2 = nn.Conv2d(nf, nf, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation)
        self.act = nn.ReLU(inplace=True)


class Encoder(nn.Module):
    """MASA feature pyramid (reference :110-143; levels 3-5 all use n_blks[2])."""

    def __init__(self, in_chl, nf, n_blks=[1, 1, 1], act='relu'):
        super().__init__()
        if act != 'relu':
            raise NotImplementedError('HIP path: Encoder uses ReLU')
        self.n_blks = list(n_blks)
        self.conv_L1 = nn.Conv2d(in_chl, nf, 3, 1, 1, bias=True)
        self.blk_L1 = make_layer(functools.partial(ResidualBlock, nf=nf), n_layers=n_blks[0])
        self.conv_L2 = nn.Conv2d(nf, nf * 2 ** 1, 3, 2, 1, bias=True)
        self.blk_L2 = make_layer(functools.partial(ResidualBlock, nf=nf * 2 ** 1), n_layers=n_blks[1])
        self.conv_L3 = nn.Conv2d(nf * 2 ** 1, nf * 2 ** 2, 3, 2, 1, bias=True)
        self.blk_L3 = make_layer(functools.partial(ResidualBlock, nf=nf * 2 ** 2), n_layers=n_blks[2])
        self.conv_L4 = nn.Conv2d(nf * 2 ** 2, nf * 2 ** 3, 3, 2, 1, bias=True)
        self.blk_L4 = make_layer(functools.partial(ResidualBlock, nf=nf * 2 ** 3), n_layers=n_blks[2])
        self.conv_L5 = nn.Conv2d(nf * 2 ** 3, nf * 2 ** 4, 3, 2, 1, bias=True)
        self.blk_L5 = make_layer(functools.partial(ResidualBlock, nf=nf * 2 ** 4), n_layers=n_blks[2])
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        names, params = _named(self)
        return list(_EncoderFn.apply(x, names, self.n_blks, *params))


class SimpleGate(nn.Module):
    """x[:, :C/2] * x[:, C/2:] -- fused into the neighbouring kernels on the HIP path;
    kept as a (parameter-free) member for module-tree parity."""

    def forward(self, x):
        raise RuntimeError('SimpleGate is fused into the depthwise / 1x1 kernels on the HIP path')


class NAFBlock(nn.Module):
    def __init__(self, c, DW_Expand=2, FFN_Expand=2, drop_out_rate=0.):
        super().__init__()
        if DW_Expand != 2 or FFN_Expand != 2 or drop_out_rate > 0.:
            raise NotImplementedError('HIP path: NAFBlock with DW_Expand=FFN_Expand=2, no dropout (reference defaults)')
        dw_channel = c * DW_Expand
        self.conv1 = nn.Conv2d(c, dw_channel, 1, padding=0, stride=1, groups=1, bias=True)
        self.conv2 = nn.Conv2d(dw_channel, dw_channel, 3, padding=1, stride=1, groups=dw_channel, bias=True)
        self.conv3 = nn.Conv2d(dw_channel // 2, c, 1, padding=0, stride=1, groups=1, bias=True)
        self.sca = nn.Sequential(nn.AdaptiveAvgPool2d(1),
                                 nn.Conv2d(dw_channel // 2, dw_channel // 2, 1, padding=0, stride=1, groups=1, bias=True))
        self.sg = SimpleGate()
        ffn_channel = FFN_Expand * c
        self.conv4 = nn.Conv2d(c, ffn_channel, 1, padding=0, stride=1, groups=1, bias=True)
        self.conv5 = nn.Conv2d(ffn_channel // 2, c, 1, padding=0, stride=1, groups=1, bias=True)
        self.norm1 = LayerNorm2d(c)
        self.norm2 = LayerNorm2d(c)
        self.dropout1 = nn.Identity()
        self.dropout2 = nn.Identity()
        self.beta = nn.Parameter(torch.zeros((1, c, 1, 1)), requires_grad=True)
        self.gamma = nn.Parameter(torch.zeros((1, c, 1, 1)), requires_grad=True)

    def forward(self, inp):
        names, params = _named(self)
        return _BlockFn.apply(inp, names, None, *params)


class NAFResFuseBlock(NAFBlock):
    """identical math on the concatenated [x, warp_ref] tensor (reference :241-302)."""


class _NAFBase(nn.Module):
    def _build_unet(self, img_channel, width, middle_blk_num, enc_blk_nums, dec_blk_nums, fusion=None):
        self.intro = nn.Conv2d(img_channel, width, 3, padding=1, stride=1, groups=1, bias=True)
        self.ending = nn.Conv2d(width, img_channel, 3, padding=1, stride=1, groups=1, bias=True)
        self.encoders = nn.ModuleList()
        self.decoders = nn.ModuleList()
        self.middle_blks = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.downs = nn.ModuleList()
        chan = width
        index = -1
        for index, num in enumerate(enc_blk_nums):
            self.encoders.append(nn.Sequential(*[NAFBlock(chan) for _ in range(num)]))
            self.downs.append(nn.Conv2d(chan, 2 * chan, 2, 2))
            if fusion is not None:
                self.masa_blk_enc.append(nn.Sequential(*[NAFResFuseBlock(chan * 2) for _ in range(fusion[index])]))
            chan = chan * 2
        self.middle_blks = nn.Sequential(*[NAFBlock(chan) for _ in range(middle_blk_num)])
        if fusion is not None:
            # reference quirk R2 (:463-465): the middle fusion count is fusion[len(enc)]
            self.masa_blk_middle.append(nn.Sequential(*[NAFResFuseBlock(chan * 2) for _ in range(fusion[index + 1])]))
        for num in dec_blk_nums:
            self.ups.append(nn.Sequential(nn.Conv2d(chan, chan * 2, 1, bias=False), nn.PixelShuffle(2)))
            chan = chan // 2
            self.decoders.append(nn.Sequential(*[NAFBlock(chan) for _ in range(num)]))
        self.padder_size = 2 ** len(self.encoders)


class NAFNet(_NAFBase):
    """the un-guided network of the same file (reference :305-386): same constructor, registration order and forward(inp)"""

    def __init__(self, img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[], dec_blk_nums=[]):
        super().__init__()
        if len(enc_blk_nums) != len(dec_blk_nums):
            raise ValueError('NAFNet: one decoder level per encoder level (the skips are zipped, reference :362-365)')
        self._build_unet(img_channel, width, middle_blk_num, enc_blk_nums, dec_blk_nums)
        self.cfg = dict(img_channel=img_channel, width=width, middle_blk_num=middle_blk_num, enc_blk_nums=list(enc_blk_nums),
                        dec_blk_nums=list(dec_blk_nums))

    def check_image_size(self, x):
        from ... import kernels as K
        _, _, h, w = x.shape
        m = self.padder_size
        return K.pad_crop(x.contiguous(), -(-h // m) * m, -(-w // m) * m)

    def forward(self, inp):
        names, params = _named(self)
        return _UNetFn.apply(inp, names, self.cfg, *params)


class NAFNetRefFusion(_NAFBase):
    def __init__(self, img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[], dec_blk_nums=[], nf=64,
                 ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[1, 1, 1, 1], reffusion_n_blocks_middle=1, scale=1,
                 num_nbr=1, psize=3, lr_block_size=8, ref_down_block_size=1.5, dilations=[1, 2, 3]):
        super().__init__()
        if nf != width:
            raise ValueError('NAFNetRefFusion needs nf == width (concat widths, reference :453,719)')
        if len(enc_blk_nums) != 4 or len(dec_blk_nums) != 4:
            raise NotImplementedError('HIP path: 4 encoder / 4 decoder levels (5-level MASA pyramid)')
        if num_nbr != 1 or psize != 3:
            raise NotImplementedError('HIP path: num_nbr=1, psize=3')
        self.scale, self.num_nbr, self.psize = scale, num_nbr, psize
        self.lr_block_size, self.ref_down_block_size, self.dilations = lr_block_size, ref_down_block_size, dilations
        self.masa_enc = Encoder(in_chl=img_channel, nf=nf, n_blks=ext_n_blocks)
        self.masa_blk_enc = nn.ModuleList()
        self.masa_blk_middle = nn.ModuleList()
        self.masa_blk_dec = nn.ModuleList()
        self._build_unet(img_channel, width, middle_blk_num, enc_blk_nums, dec_blk_nums, fusion=reffusion_n_blocks)
        self.cfg = dict(img_channel=img_channel, width=width, middle_blk_num=middle_blk_num,
                        enc_blk_nums=list(enc_blk_nums), dec_blk_nums=list(dec_blk_nums), nf=nf,
                        ext_n_blocks=list(ext_n_blocks), reffusion_n_blocks=list(reffusion_n_blocks),
                        lr_block_size=lr_block_size, ref_down_block_size=ref_down_block_size,
                        dilations=list(dilations), psize=psize)

    def check_image_size(self, x):
        mult = self.padder_size * self.lr_block_size
        _, _, h, w = x.shape
        from ... import kernels as K
        return K.pad_crop(x.contiguous(), -(-h // mult) * mult, -(-w // mult) * mult)

    def forward(self, inp, ref):
        names, params = _named(self)
        return _NetFn.apply(inp, ref, names, self.cfg, *params)


class NAFNetLocal(NAFNet):
    """TLSC test-time wrapper of the un-guided NAFNet (reference :756-768 + models/archs/nafnet_local_arch.py:10-104): same
    constructor (`train_size`, `fast_imp`), same parameters / state dict as NAFNet; every NAFBlock's global average pool becomes a
    local box mean whose kernel is fixed at construction -- 1.5 x the feature size the network sees at `train_size`
    (engine.tlsc_kernel_sizes restates what `Local_Base.convert`'s first forward computes; that forward itself only serves to fix
    the kernels there).  Inference only: the module is put in eval mode and its forward runs without autograd, as in the reference.
    `fast_imp=True` selects the reference's sub-sampled "non-equivalent but faster" variant (:46-60), not built here."""

    def __init__(self, *args, train_size=(1, 3, 256, 256), fast_imp=False, **kwargs):
        super().__init__(*args, **kwargs)
        if fast_imp:
            raise NotImplementedError('NAFNetLocal(fast_imp=True): the sub-sampled integral-image variant is not built (fast_imp=False '
                                      'is what every shipped configuration uses)')
        self.train_size, self.fast_imp = tuple(train_size), fast_imp
        self.ksizes = E.tlsc_kernel_sizes(self.cfg, self.train_size)
        self.eval()

    def forward(self, inp):
        require_gpu(inp, 'NAFNetLocal')
        names, params = _named(self)
        with torch.no_grad():
            out, _ = E.unet_fwd(dict(zip(names, [p.detach() for p in params])), self.cfg, inp.detach(), local=self.ksizes)
        return out


class NAFNetLocal_RefFusion(NAFNetRefFusion):
    """The reference's TLSC test-time wrapper of the guided NAFNet (network_nafnet_guided_arch.py:743-753) cannot be
    constructed there: Local_Base.convert (nafnet_local_arch.py:99-104) runs `self.forward(imgs)` with one argument, and the
    guided forward needs (inp, ref) -- defect R7, recorded by tests/golden/make_golden_defects.py.  Same error here."""

    def __init__(self, *args, train_size=(1, 3, 256, 256), fast_imp=False, **kwargs):
        raise TypeError("NAFNetRefFusion.forward() missing 1 required positional argument: 'ref' "
                        "(NAFNetLocal_RefFusion cannot be constructed in the reference either: defect R7)")


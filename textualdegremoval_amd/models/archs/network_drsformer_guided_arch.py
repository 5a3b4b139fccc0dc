"""DRSformer guided architecture (with the MEFC sub-networks) on the HIP engine.

Drop-in mirror of `DRSformerRefFusion` (reference models/archs/network_drsformer_guided_arch.py:679-1123, the network of
008/009/010_drsformer_image_deraining_*.yml): same constructor kwargs, forward signature, parameter names, registration
order (= state-dict order) and default initialisation.  The nn.Conv2d / nn.Linear members are parameter containers only --
all arithmetic runs in libtdr_hip.so through textualdegremoval_amd.drsformer_engine (cfg['mefc'] = True).

Reference defect R1 (the 4-entry encoder pyramid indexed at feat[1..4]) applies as in the other guided transformer files;
this file does import functools and its level-1 fusion is wired correctly (R5 / R6 are defects of the 200L_SPA file only).
"""
import torch
import torch.nn as nn

from ... import drsformer_engine as DE
from ... import kernels as K
from .nafnet_arch_utils import require_gpu
from .network_drsformer_guided_200L_SPA_arch import TransformerBlock, TransformerResFusionBlock
from .network_restormer_guided_arch import Downsample, Encoder, OverlapPatchEmbed, Upsample, _named  # noqa: F401

Operations = ['sep_conv_1x1', 'sep_conv_3x3', 'sep_conv_5x5', 'sep_conv_7x7', 'dil_conv_3x3', 'dil_conv_5x5', 'dil_conv_7x7',
              'avg_pool_3x3']


class ReLUConv(nn.Module):          # (:466-474: despite the name, conv THEN relu)
    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        super().__init__()
        self.op = nn.Sequential(nn.Conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False), nn.ReLU(inplace=False))


class DilConv(nn.Module):           # (:477-486)
    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.Conv2d(C_in, C_in, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, groups=C_in, bias=False),
            nn.Conv2d(C_in, C_out, kernel_size=1, padding=0, bias=False), )


class SepConv(nn.Module):           # (:507-519)
    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.Conv2d(C_in, C_in, kernel_size=kernel_size, stride=stride, padding=padding, groups=C_in, bias=False),
            nn.Conv2d(C_in, C_in, kernel_size=1, padding=0, bias=False),
            nn.ReLU(inplace=False),
            nn.Conv2d(C_in, C_in, kernel_size=kernel_size, stride=1, padding=padding, groups=C_in, bias=False),
            nn.Conv2d(C_in, C_out, kernel_size=1, padding=0, bias=False), )


OPS = {
    'avg_pool_3x3': lambda C, stride, affine: nn.AvgPool2d(3, stride=stride, padding=1, count_include_pad=False),
    'sep_conv_1x1': lambda C, stride, affine: SepConv(C, C, 1, stride, 0, affine=affine),
    'sep_conv_3x3': lambda C, stride, affine: SepConv(C, C, 3, stride, 1, affine=affine),
    'sep_conv_5x5': lambda C, stride, affine: SepConv(C, C, 5, stride, 2, affine=affine),
    'sep_conv_7x7': lambda C, stride, affine: SepConv(C, C, 7, stride, 3, affine=affine),
    'dil_conv_3x3': lambda C, stride, affine: DilConv(C, C, 3, stride, 2, 2, affine=affine),
    'dil_conv_5x5': lambda C, stride, affine: DilConv(C, C, 5, stride, 4, 2, affine=affine),
    'dil_conv_7x7': lambda C, stride, affine: DilConv(C, C, 7, stride, 6, 2, affine=affine),
}


class OperationLayer(nn.Module):    # (:371-386)
    def __init__(self, C, stride):
        super().__init__()
        self._ops = nn.ModuleList()
        for o in Operations:
            self._ops.append(OPS[o](C, stride, False))
        self._out = nn.Sequential(nn.Conv2d(C * len(Operations), C, 1, padding=0, bias=False), nn.ReLU())


class GroupOLs(nn.Module):          # (:389-408)
    def __init__(self, steps, C):
        super().__init__()
        self.preprocess = ReLUConv(C, C, 1, 1, 0, affine=False)
        self._steps = steps
        self._ops = nn.ModuleList()
        self.relu = nn.ReLU()
        for _ in range(self._steps):
            self._ops.append(OperationLayer(C, 1))


class OALayer(nn.Module):           # (:411-428)
    def __init__(self, channel, k, num_ops):
        super().__init__()
        self.k = k
        self.num_ops = num_ops
        self.output = k * num_ops
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.ca_fc = nn.Sequential(nn.Linear(channel, self.output * 2), nn.ReLU(), nn.Linear(self.output * 2, self.k * self.num_ops))


class _SubnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, names, *params):
        require_gpu(x, 'subnet')
        P = {'s.' + k: p.detach() for k, p in zip(names, params)}
        out, saved = DE.mefc_fwd(x.contiguous(), P, 's.')
        ctx.names, ctx.P, ctx.saved = names, P, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = {}
        dx = DE.mefc_bwd(dout.contiguous(), ctx.P, 's.', ctx.saved, G)
        return (dx, None) + tuple(G['s.' + k].view_as(ctx.P['s.' + k]) for k in ctx.names)


class subnet(nn.Module):
    """Mixture of Experts Feature Compensator (:522-548)."""

    def __init__(self, dim, layer_num=1, steps=4):
        super().__init__()
        if layer_num != 1 or steps != 4:
            raise NotImplementedError('HIP path: subnet(layer_num=1, steps=4), the only form the reference instantiates')
        self._C = dim
        self.num_ops = len(Operations)
        self._layer_num = layer_num
        self._steps = steps
        self.layers = nn.ModuleList()
        for _ in range(self._layer_num):
            self.layers += [OALayer(self._C, self._steps, self.num_ops)]
            self.layers += [GroupOLs(steps, self._C)]

    def forward(self, x):
        names, params = _named(self)
        return _SubnetFn.apply(x, names, *params)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, ref, names, cfg, *params):
        require_gpu(inp, 'DRSformerRefFusion')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = DE.net_fwd(P, cfg, inp, ref)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = DE.net_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[k].view_as(ctx.P[k]) for k in ctx.names)


class DRSformer(nn.Module):
    """the un-guided network of the same file (reference :586-676): same constructor, registration order and `forward(inp_img)`;
    MEFC sub-networks before the encoder and as the refinement stage; runs on the guided engine without the reference branch."""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], heads=[1, 2, 4, 8],
                 ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias'):
        super().__init__()
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level0 = subnet(dim)

        def blocks(n, c, h):
            return nn.Sequential(*[TransformerBlock(dim=c, num_heads=h, ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                                    LayerNorm_type=LayerNorm_type) for _ in range(n)])
        self.encoder_level1 = blocks(num_blocks[0], dim, heads[0])
        self.down1_2 = Downsample(dim)
        self.encoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.encoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.latent = blocks(num_blocks[3], int(dim * 2 ** 3), heads[3])
        self.up4_3 = Upsample(int(dim * 2 ** 3))
        self.reduce_chan_level3 = nn.Conv2d(int(dim * 2 ** 3), int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = nn.Conv2d(int(dim * 2 ** 2), int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.decoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.decoder_level1 = blocks(num_blocks[0], int(dim * 2 ** 1), heads[0])
        self.refinement = subnet(dim=int(dim * 2 ** 1))
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks), heads=list(heads),
                        ffn_expansion_factor=ffn_expansion_factor, bias=bias, LayerNorm_type=LayerNorm_type, mefc=True)

    def forward(self, inp_img):
        names, params = _named(self)
        return _NetFn.apply(inp_img, None, names, self.cfg, *params)


class DRSformerRefFusion(nn.Module):
    engine = DE

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], heads=[1, 2, 4, 8],
                 ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', nf=64, ext_n_blocks=[4, 4, 4, 4],
                 reffusion_n_blocks=[1, 1, 1, 1], reffusion_n_blocks_middle=1, scale=1, num_nbr=1, psize=3, lr_block_size=8,
                 ref_down_block_size=1.5, dilations=[1, 2, 3]):
        super().__init__()
        if nf != dim:
            raise ValueError('DRSformerRefFusion needs nf == dim (the fusion blocks are built for 2*dim*2^l channels)')
        if num_nbr != 1 or psize != 3:
            raise NotImplementedError('HIP path: num_nbr=1, psize=3')
        self.scale, self.num_nbr, self.psize = scale, num_nbr, psize
        self.lr_block_size, self.ref_down_block_size, self.dilations = lr_block_size, ref_down_block_size, dilations
        self.padder_size = 2 ** 3
        self.masa_enc = Encoder(in_chl=inp_channels, nf=nf, n_blks=ext_n_blocks)
        self.masa_blk_enc = nn.ModuleList()
        self.masa_blk_middle = nn.ModuleList()
        self.masa_blk_dec = nn.ModuleList()
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level0 = subnet(dim)

        def blocks(n, c, h, cls=TransformerBlock):
            return nn.Sequential(*[cls(dim=c, num_heads=h, ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                       LayerNorm_type=LayerNorm_type) for _ in range(n)])
        F = TransformerResFusionBlock
        self.masa_blk_enc_level1 = blocks(reffusion_n_blocks[0], 2 * dim, heads[0], F)
        self.encoder_level1 = blocks(num_blocks[0], dim, heads[0])
        self.down1_2 = Downsample(dim)
        self.masa_blk_enc_level2 = blocks(reffusion_n_blocks[1], 2 * dim * 2 ** 1, heads[1], F)
        self.encoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.masa_blk_enc_level3 = blocks(reffusion_n_blocks[2], 2 * dim * 2 ** 2, heads[2], F)
        self.encoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.masa_blk_enc_level4 = blocks(reffusion_n_blocks[3], 2 * dim * 2 ** 3, heads[3], F)
        self.latent = blocks(num_blocks[3], int(dim * 2 ** 3), heads[3])
        self.up4_3 = Upsample(int(dim * 2 ** 3))
        self.reduce_chan_level3 = nn.Conv2d(int(dim * 2 ** 3), int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = nn.Conv2d(int(dim * 2 ** 2), int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.decoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.decoder_level1 = blocks(num_blocks[0], int(dim * 2 ** 1), heads[0])
        self.refinement = subnet(dim=int(dim * 2 ** 1))
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks),
                        heads=list(heads), ffn_expansion_factor=ffn_expansion_factor, bias=bias, LayerNorm_type=LayerNorm_type,
                        nf=nf, ext_n_blocks=list(ext_n_blocks), reffusion_n_blocks=list(reffusion_n_blocks),
                        lr_block_size=lr_block_size, ref_down_block_size=ref_down_block_size, dilations=list(dilations), psize=psize,
                        mefc=True)

    def check_image_size(self, x):
        mult = self.padder_size * self.lr_block_size
        _, _, h, w = x.shape
        return K.pad_crop(x.contiguous(), -(-h // mult) * mult, -(-w // mult) * mult)

    def forward(self, inp_img, ref_img):
        names, params = _named(self)
        return _NetFn.apply(inp_img, ref_img, names, self.cfg, *params)

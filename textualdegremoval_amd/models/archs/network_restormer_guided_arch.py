"""Restormer guided architecture on the HIP engine.

Drop-in mirror of the reference's models/archs/network_restormer_guided_arch.py: same class names,
constructor kwargs, forward signatures, parameter names, registration order (= state-dict order, optimizer
group order) and default initialisation (the nn.Conv2d members are constructed exactly like the
reference's, so the RNG stream and init match; they are parameter containers only -- their ATen forward is
never called).  All arithmetic runs in libtdr_hip.so through textualdegremoval_amd.restormer_engine.
"""
import functools
import numbers

import torch
import torch.nn as nn

from ... import engine as E
from ... import kernels as K
from ... import restormer_engine as R
from .nafnet_arch_utils import require_gpu


def _named(module):
    names, params = [], []
    for k, p in module.named_parameters():
        names.append(k)
        params.append(p)
    return names, params


# ---------------------------------------------------------------------------- LayerNorm (:172-218)
class _LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        require_gpu(x, 'LayerNorm')
        x = x.contiguous()
        center = bias is not None
        y, mu, rstd = K.layernorm2d_fwd(x, weight, bias, R.LN_EPS, center=center)
        ctx.save_for_backward(x, mu, rstd, weight)
        ctx.center = center
        return y

    @staticmethod
    def backward(ctx, go):
        x, mu, rstd, weight = ctx.saved_tensors
        gx, gw, gb = K.layernorm2d_bwd(go.contiguous(), x, mu, rstd, weight, center=ctx.center)
        return gx, gw, (gb if ctx.center else None)


def _norm_shape(normalized_shape):
    if isinstance(normalized_shape, numbers.Integral):
        normalized_shape = (normalized_shape,)
    normalized_shape = torch.Size(normalized_shape)
    assert len(normalized_shape) == 1
    return normalized_shape


class BiasFree_LayerNorm(nn.Module):
    """x / sqrt(var + 1e-5) * weight over the channel dim of an NCHW tensor (the reference's to_3d / to_4d
    round trip is a layout change only)."""

    def __init__(self, normalized_shape):
        super().__init__()
        self.normalized_shape = _norm_shape(normalized_shape)
        self.weight = nn.Parameter(torch.ones(self.normalized_shape))

    def forward(self, x):
        return _LNFn.apply(x, self.weight, None)


class WithBias_LayerNorm(nn.Module):
    def __init__(self, normalized_shape):
        super().__init__()
        self.normalized_shape = _norm_shape(normalized_shape)
        self.weight = nn.Parameter(torch.ones(self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(self.normalized_shape))

    def forward(self, x):
        return _LNFn.apply(x, self.weight, self.bias)


class LayerNorm(nn.Module):
    def __init__(self, dim, LayerNorm_type):
        super().__init__()
        self.body = BiasFree_LayerNorm(dim) if LayerNorm_type == 'BiasFree' else WithBias_LayerNorm(dim)

    def forward(self, x):
        return self.body(x)


# ---------------------------------------------------------------------------- blocks (:223-353)
class FeedForward(nn.Module):
    """GDFN; parameter container (its math is fused into the TransformerBlock node)."""

    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden_features = int(dim * ffn_expansion_factor)
        self.project_in = nn.Conv2d(dim, hidden_features * 2, kernel_size=1, bias=bias)
        self.dwconv = nn.Conv2d(hidden_features * 2, hidden_features * 2, kernel_size=3, stride=1, padding=1,
                                groups=hidden_features * 2, bias=bias)
        self.project_out = nn.Conv2d(hidden_features, dim, kernel_size=1, bias=bias)


class Attention(nn.Module):
    """MDTA; parameter container (its math is fused into the TransformerBlock node)."""

    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, kernel_size=1, bias=bias)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, names, heads, ln_type, fusion, *params):
        require_gpu(x, 'TransformerBlock')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = (R.fblock_fwd if fusion else R.tblock_fwd)(x.contiguous(), P, heads, ln_type)
        ctx.names, ctx.P, ctx.saved, ctx.meta = names, P, saved, (heads, ln_type, fusion)
        return out

    @staticmethod
    def backward(ctx, dout):
        heads, ln_type, fusion = ctx.meta
        dx, G = (R.fblock_bwd if fusion else R.tblock_bwd)(dout.contiguous(), ctx.P, heads, ln_type, ctx.saved)
        return (dx, None, None, None, None) + tuple(G[k] for k in ctx.names)


class TransformerBlock(nn.Module):
    _fusion = False

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = Attention(dim, num_heads, bias)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)
        self._heads, self._ln = num_heads, LayerNorm_type

    def forward(self, x):
        names, params = _named(self)
        return _BlockFn.apply(x, names, self._heads, self._ln, self._fusion, *params)


class TransformerResFusionBlock(TransformerBlock):
    """block(x) * alpha + x (:334-353); alpha is registered after the sub-modules are declared but, being a
    Parameter of the block itself, comes first in named_parameters() -- same as the reference."""
    _fusion = True

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__(dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type)
        self.alpha = nn.Parameter(torch.zeros(1), requires_grad=True)


# ---------------------------------------------------------------------------- resizing / embedding (:358-391)
class OverlapPatchEmbed(nn.Module):
    def __init__(self, in_c=3, embed_dim=48, bias=False):
        super().__init__()
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=3, stride=1, padding=1, bias=bias)


class Downsample(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat // 2, kernel_size=3, stride=1, padding=1, bias=False),
                                  nn.PixelUnshuffle(2))


class Upsample(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat * 2, kernel_size=3, stride=1, padding=1, bias=False),
                                  nn.PixelShuffle(2))


# ---------------------------------------------------------------------------- MASA encoder (:44-59, :99-133)
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
    """the 4-level pyramid of this file (levels 3,4 both use n_blks[2])."""

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
        self.act = nn.ReLU(inplace=True)


# ---------------------------------------------------------------------------- whole network (:504-963)
class _NetFn(torch.autograd.Function):
    """whole RestormerRefFusion forward/backward in one node (every parameter is used exactly once)."""

    @staticmethod
    def forward(ctx, inp, ref, names, cfg, *params):
        require_gpu(inp, 'RestormerRefFusion')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = R.net_fwd(P, cfg, inp, ref)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = R.net_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[k] for k in ctx.names)


class _UNetFn(torch.autograd.Function):
    """the un-guided Restormer as one autograd node"""

    @staticmethod
    def forward(ctx, inp, names, cfg, *params):
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = R.unet_fwd(P, cfg, inp)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = R.unet_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (None, None, None) + tuple(G[k] for k in ctx.names)


class Restormer(nn.Module):
    """the un-guided network of the same file (reference :396-501): same constructor, registration order, forward(inp_img)"""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, heads=[1, 2, 4, 8],
                 ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', dual_pixel_task=False):
        super().__init__()
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)

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
        self.refinement = blocks(num_refinement_blocks, int(dim * 2 ** 1), heads[0])
        self.dual_pixel_task = dual_pixel_task
        if self.dual_pixel_task:
            self.skip_conv = nn.Conv2d(dim, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks),
                        num_refinement_blocks=num_refinement_blocks, heads=list(heads), ffn_expansion_factor=ffn_expansion_factor,
                        bias=bias, LayerNorm_type=LayerNorm_type, dual_pixel_task=dual_pixel_task)

    def forward(self, inp_img):
        names, params = _named(self)
        return _UNetFn.apply(inp_img, names, self.cfg, *params)


class RestormerRefFusion(nn.Module):
    engine = R          # image_restoration_ref_model dispatches its fused step through `net.engine`

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias',
                 dual_pixel_task=False, nf=64, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[1, 1, 1, 1],
                 reffusion_n_blocks_middle=1, scale=1, num_nbr=1, psize=3, lr_block_size=8, ref_down_block_size=1.5,
                 dilations=[1, 2, 3]):
        super().__init__()
        if nf != dim:
            raise ValueError('RestormerRefFusion needs nf == dim (the fusion blocks are built for 2*dim*2^l channels, '
                             'reference :563,583,603,623)')
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
        self.refinement = blocks(num_refinement_blocks, int(dim * 2 ** 1), heads[0])
        self.dual_pixel_task = dual_pixel_task
        if self.dual_pixel_task:           # dual-pixel defocus deblurring (:634-637)
            self.skip_conv = nn.Conv2d(dim, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks),
                        num_refinement_blocks=num_refinement_blocks, heads=list(heads),
                        ffn_expansion_factor=ffn_expansion_factor, bias=bias, LayerNorm_type=LayerNorm_type,
                        dual_pixel_task=dual_pixel_task, nf=nf, ext_n_blocks=list(ext_n_blocks),
                        reffusion_n_blocks=list(reffusion_n_blocks), lr_block_size=lr_block_size,
                        ref_down_block_size=ref_down_block_size, dilations=list(dilations), psize=psize)

    def check_image_size(self, x):
        mult = self.padder_size * self.lr_block_size
        _, _, h, w = x.shape
        return K.pad_crop(x.contiguous(), -(-h // mult) * mult, -(-w // mult) * mult)

    def forward(self, inp_img, ref_img):
        names, params = _named(self)
        return _NetFn.apply(inp_img, ref_img, names, self.cfg, *params)

"""DRSformer guided architecture without the MEFC sub-network, on the HIP engine.

Drop-in mirror of `DRSformer200L_SPA_RefFusion` (reference models/archs/network_drsformer_guided_arch_200L_SPA.py:582-, the
network of 007_drsformer_image_deraining_rain200l.yml): same constructor kwargs, forward signature, parameter names,
registration order (= state-dict order) and default initialisation.  nn.Conv2d members are parameter containers only -- all
arithmetic runs in libtdr_hip.so through textualdegremoval_amd.drsformer_engine.

Reference defects (oracle/drsformer_ref_oracle.py): the reference class cannot be constructed as shipped (R5: functools is
never imported) and its forward indexes the encoder pyramid one slot off (R1); with both repaired, its level-1 reference
fusion is computed and discarded (R6) -- `masa_blk_enc_level1.*` are part of the state dict and never receive a gradient.
"""
import torch
import torch.nn as nn

from ... import drsformer_engine as DE
from ... import kernels as K
from .nafnet_arch_utils import require_gpu
from .network_restormer_guided_arch import Downsample, Encoder, LayerNorm, OverlapPatchEmbed, Upsample, _named  # noqa: F401


class FeedForward(nn.Module):
    """mixed-scale feed-forward (:213-253); parameter container."""

    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden_features = int(dim * ffn_expansion_factor)
        self.project_in = nn.Conv2d(dim, hidden_features * 2, kernel_size=1, bias=bias)
        self.dwconv3x3 = nn.Conv2d(hidden_features * 2, hidden_features * 2, kernel_size=3, stride=1, padding=1,
                                   groups=hidden_features * 2, bias=bias)
        self.dwconv5x5 = nn.Conv2d(hidden_features * 2, hidden_features * 2, kernel_size=5, stride=1, padding=2,
                                   groups=hidden_features * 2, bias=bias)
        self.relu3 = nn.ReLU()
        self.relu5 = nn.ReLU()
        self.dwconv3x3_1 = nn.Conv2d(hidden_features * 2, hidden_features, kernel_size=3, stride=1, padding=1,
                                     groups=hidden_features, bias=bias)
        self.dwconv5x5_1 = nn.Conv2d(hidden_features * 2, hidden_features, kernel_size=5, stride=1, padding=2,
                                     groups=hidden_features, bias=bias)
        self.relu3_1 = nn.ReLU()
        self.relu5_1 = nn.ReLU()
        self.project_out = nn.Conv2d(hidden_features * 2, dim, kernel_size=1, bias=bias)


class Attention(nn.Module):
    """Top-K Sparse Attention (:257-328); parameter container."""

    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, kernel_size=1, bias=bias)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.attn_drop = nn.Dropout(0.)
        self.attn1 = torch.nn.Parameter(torch.tensor([0.2]), requires_grad=True)
        self.attn2 = torch.nn.Parameter(torch.tensor([0.2]), requires_grad=True)
        self.attn3 = torch.nn.Parameter(torch.tensor([0.2]), requires_grad=True)
        self.attn4 = torch.nn.Parameter(torch.tensor([0.2]), requires_grad=True)


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, names, heads, ln_type, fusion, *params):
        require_gpu(x, 'TransformerBlock')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = (DE.fblock_fwd if fusion else DE.tblock_fwd)(x.contiguous(), P, heads, ln_type)
        ctx.names, ctx.P, ctx.saved, ctx.meta = names, P, saved, (heads, ln_type, fusion)
        return out

    @staticmethod
    def backward(ctx, dout):
        heads, ln_type, fusion = ctx.meta
        dx, G = (DE.fblock_bwd if fusion else DE.tblock_bwd)(dout.contiguous(), ctx.P, heads, ln_type, ctx.saved)
        return (dx, None, None, None, None) + tuple(G[k].view_as(ctx.P[k]) for k in ctx.names)


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
    _fusion = True

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__(dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type)
        self.alpha = nn.Parameter(torch.zeros(1), requires_grad=True)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, ref, names, cfg, *params):
        require_gpu(inp, 'DRSformer200L_SPA_RefFusion')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = DE.net_fwd(P, cfg, inp, ref)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = DE.net_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[k].view_as(ctx.P[k]) for k in ctx.names)


class DRSformer200L_SPA_RefFusion(nn.Module):
    engine = DE
    unused_parameter_prefixes = ('masa_blk_enc_level1.',)      # R6: computed and discarded by the reference, no gradient

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], heads=[1, 2, 4, 8],
                 ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', nf=64, ext_n_blocks=[4, 4, 4, 4],
                 reffusion_n_blocks=[1, 1, 1, 1], reffusion_n_blocks_middle=1, scale=1, num_nbr=1, psize=3, lr_block_size=8,
                 ref_down_block_size=1.5, dilations=[1, 2, 3]):
        super().__init__()
        if nf != dim:
            raise ValueError('DRSformer200L_SPA_RefFusion needs nf == dim (the fusion blocks are built for 2*dim*2^l channels)')
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
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks),
                        heads=list(heads), ffn_expansion_factor=ffn_expansion_factor, bias=bias, LayerNorm_type=LayerNorm_type,
                        nf=nf, ext_n_blocks=list(ext_n_blocks), reffusion_n_blocks=list(reffusion_n_blocks),
                        lr_block_size=lr_block_size, ref_down_block_size=ref_down_block_size, dilations=list(dilations), psize=psize)
        for k, p in self.named_parameters():
            if k.startswith(self.unused_parameter_prefixes):
                p.requires_grad_(False)

    def used_named_parameters(self):
        return [(k, p) for k, p in self.named_parameters() if not k.startswith(self.unused_parameter_prefixes)]

    def check_image_size(self, x):
        mult = self.padder_size * self.lr_block_size
        _, _, h, w = x.shape
        return K.pad_crop(x.contiguous(), -(-h // mult) * mult, -(-w // mult) * mult)

    def forward(self, inp_img, ref_img):
        names, params = zip(*self.used_named_parameters())
        return _NetFn.apply(inp_img, ref_img, list(names), self.cfg, *params)

"""PromptIR guided architecture on the HIP engine.

Drop-in mirror of the reference's models/archs/network_promptir_guided_arch.py for the class the trainer instantiates
(`PromptIRRefFusion`, 001_promptir_all_in_one_restoration.yml:48): same constructor kwargs, forward signature, parameter
names, registration order (= state-dict order) and default initialisation.  The nn.Conv2d / nn.Linear members are
parameter containers only -- all arithmetic runs in libtdr_hip.so through textualdegremoval_amd.promptir_engine.

The shared building blocks (LayerNorm, Attention, FeedForward, TransformerBlock, TransformerResFusionBlock, Downsample,
Upsample, OverlapPatchEmbed, Encoder, ResidualBlock) are identical in the reference's two files and are the mirrors of
network_restormer_guided_arch.py.

What the reference allows (oracle/promptir_ref_oracle.py, defects R1 / R4): the network runs only with `decoder=True`
and dim = nf = 48 -- with the YAML's `decoder: False` the reference raises a RuntimeError in up4_3 on its first forward
pass.  This class raises a ValueError naming that defect at construction instead.
"""
import torch
import torch.nn as nn

from ... import kernels as K
from ... import promptir_engine as PE
from .nafnet_arch_utils import require_gpu
from .network_restormer_guided_arch import (Attention, BiasFree_LayerNorm, Downsample, Encoder, FeedForward,  # noqa: F401
                                            LayerNorm, OverlapPatchEmbed, ResidualBlock, TransformerBlock,
                                            TransformerResFusionBlock, Upsample, WithBias_LayerNorm, _named, make_layer)


class _PromptFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, names, *params):
        require_gpu(x, 'PromptGenBlock')
        P = {'p.' + k: p.detach() for k, p in zip(names, params)}
        out, saved = PE.prompt_fwd(x.contiguous(), P, 'p.')
        ctx.names, ctx.P, ctx.saved = names, P, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = {}
        demb, inv = PE.prompt_bwd(dout.contiguous(), ctx.P, 'p.', ctx.saved, G)
        N, C, H, W = ctx.saved[0]
        dx = K.plane_add_(torch.zeros(N, C, H, W, dtype=torch.float32, device=dout.device), demb, inv)
        return (dx, None) + tuple(G['p.' + k] for k in ctx.names)


class PromptGenBlock(nn.Module):
    """:417-441 -- prompt = conv3x3(resize(sum_k softmax(Linear(mean_hw x))_k * prompt_param_k))."""

    def __init__(self, prompt_dim=128, prompt_len=5, prompt_size=96, lin_dim=192):
        super().__init__()
        self.prompt_param = nn.Parameter(torch.rand(1, prompt_len, prompt_dim, prompt_size, prompt_size))
        self.linear_layer = nn.Linear(lin_dim, prompt_len)
        self.conv3x3 = nn.Conv2d(prompt_dim, prompt_dim, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        names, params = _named(self)
        return _PromptFn.apply(x, names, *params)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, ref, names, cfg, *params):
        require_gpu(inp, 'PromptIRRefFusion')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = PE.net_fwd(P, cfg, inp, ref)
        ctx.names, ctx.P, ctx.cfg, ctx.saved = names, P, cfg, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = PE.net_bwd(dout, ctx.P, ctx.cfg, ctx.saved)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[k] for k in ctx.names)


class PromptIR(nn.Module):
    """the un-guided network of the same file (reference :443-590): same constructor, registration order, parameter names and
    `forward(inp_img, noise_emb=None)`; runs on the guided engine without the reference branch (promptir_engine.net_fwd(ref=None)).
    Like the guided class it only runs as decoder=True with dim = 48 (R4); `chnl_reduce1-3` / `reduce_noise_channel_1-3` are
    registered and never used, as in the reference."""
    unused_parameter_prefixes = ('chnl_reduce1.', 'chnl_reduce2.', 'chnl_reduce3.', 'reduce_noise_channel_1.',
                                 'reduce_noise_channel_2.', 'reduce_noise_channel_3.')

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', decoder=False):
        super().__init__()
        if not decoder:
            raise ValueError('PromptIR(decoder=False): the reference raises in up4_3 on its first forward pass (384-channel latent into '
                             'Upsample(dim*4), network_promptir_guided_arch.py:497,557); only decoder=True runs')
        if dim != 48:
            raise ValueError('PromptIR(decoder=True) needs dim = 48: the prompt widths 64/128/320 and the +192/+224/+512 channel counts '
                             'are hard-wired in the reference (:463-465, :498-521)')
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.decoder = decoder
        self.prompt1 = PromptGenBlock(prompt_dim=64, prompt_len=5, prompt_size=64, lin_dim=96)
        self.prompt2 = PromptGenBlock(prompt_dim=128, prompt_len=5, prompt_size=32, lin_dim=192)
        self.prompt3 = PromptGenBlock(prompt_dim=320, prompt_len=5, prompt_size=16, lin_dim=384)
        self.chnl_reduce1 = nn.Conv2d(64, 64, kernel_size=1, bias=bias)
        self.chnl_reduce2 = nn.Conv2d(128, 128, kernel_size=1, bias=bias)
        self.chnl_reduce3 = nn.Conv2d(320, 256, kernel_size=1, bias=bias)
        self.reduce_noise_channel_1 = nn.Conv2d(dim + 64, dim, kernel_size=1, bias=bias)

        def blocks(n, c, h):
            return nn.Sequential(*[TransformerBlock(dim=c, num_heads=h, ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                                    LayerNorm_type=LayerNorm_type) for _ in range(n)])

        def tblock(c, h):
            return TransformerBlock(dim=c, num_heads=h, ffn_expansion_factor=ffn_expansion_factor, bias=bias, LayerNorm_type=LayerNorm_type)
        self.encoder_level1 = blocks(num_blocks[0], dim, heads[0])
        self.down1_2 = Downsample(dim)
        self.reduce_noise_channel_2 = nn.Conv2d(int(dim * 2 ** 1) + 128, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.encoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.reduce_noise_channel_3 = nn.Conv2d(int(dim * 2 ** 2) + 256, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.encoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.latent = blocks(num_blocks[3], int(dim * 2 ** 3), heads[3])
        self.up4_3 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level3 = nn.Conv2d(int(dim * 2 ** 1) + 192, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.noise_level3 = tblock(int(dim * 2 ** 2) + 512, heads[2])
        self.reduce_noise_level3 = nn.Conv2d(int(dim * 2 ** 2) + 512, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = nn.Conv2d(int(dim * 2 ** 2), int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.noise_level2 = tblock(int(dim * 2 ** 1) + 224, heads[2])
        self.reduce_noise_level2 = nn.Conv2d(int(dim * 2 ** 1) + 224, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.noise_level1 = tblock(int(dim * 2 ** 1) + 64, heads[2])
        self.reduce_noise_level1 = nn.Conv2d(int(dim * 2 ** 1) + 64, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.decoder_level1 = blocks(num_blocks[0], int(dim * 2 ** 1), heads[0])
        self.refinement = blocks(num_refinement_blocks, int(dim * 2 ** 1), heads[0])
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks),
                        num_refinement_blocks=num_refinement_blocks, heads=list(heads), ffn_expansion_factor=ffn_expansion_factor,
                        bias=bias, LayerNorm_type=LayerNorm_type, decoder=decoder)
        for k, p in self.named_parameters():
            if k.startswith(self.unused_parameter_prefixes):
                p.requires_grad_(False)

    def used_named_parameters(self):
        return [(k, p) for k, p in self.named_parameters() if not k.startswith(self.unused_parameter_prefixes)]

    def forward(self, inp_img, noise_emb=None):
        names, params = zip(*self.used_named_parameters())
        return _NetFn.apply(inp_img, None, list(names), self.cfg, *params)


class PromptIRRefFusion(nn.Module):
    engine = PE         # image_restoration_ref_model dispatches its fused step through `net.engine`
    # registered by the reference (:647-651, :668, :687) but never used in forward: part of the state dict, no gradient
    unused_parameter_prefixes = ('chnl_reduce1.', 'chnl_reduce2.', 'chnl_reduce3.', 'reduce_noise_channel_1.',
                                 'reduce_noise_channel_2.', 'reduce_noise_channel_3.')

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', decoder=False,
                 nf=64, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[1, 1, 1, 1], reffusion_n_blocks_middle=1, scale=1,
                 num_nbr=1, psize=3, lr_block_size=8, ref_down_block_size=1.5, dilations=[1, 2, 3]):
        super().__init__()
        if not decoder:
            raise ValueError('PromptIRRefFusion(decoder=False): the reference raises in up4_3 on its first forward pass '
                             '(384-channel latent into Upsample(dim*4), network_promptir_guided_arch.py:733,1065); '
                             'only decoder=True runs')
        if dim != 48 or nf != 48:
            raise ValueError('PromptIRRefFusion(decoder=True) needs dim = nf = 48: the prompt widths 64/128/320 and the '
                             '+192/+224/+512 channel counts are hard-wired in the reference (:643-645, :734-757)')
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
        self.decoder = decoder
        self.prompt1 = PromptGenBlock(prompt_dim=64, prompt_len=5, prompt_size=64, lin_dim=96)
        self.prompt2 = PromptGenBlock(prompt_dim=128, prompt_len=5, prompt_size=32, lin_dim=192)
        self.prompt3 = PromptGenBlock(prompt_dim=320, prompt_len=5, prompt_size=16, lin_dim=384)
        self.chnl_reduce1 = nn.Conv2d(64, 64, kernel_size=1, bias=bias)
        self.chnl_reduce2 = nn.Conv2d(128, 128, kernel_size=1, bias=bias)
        self.chnl_reduce3 = nn.Conv2d(320, 256, kernel_size=1, bias=bias)
        self.reduce_noise_channel_1 = nn.Conv2d(dim + 64, dim, kernel_size=1, bias=bias)

        def blocks(n, c, h, cls=TransformerBlock):
            return nn.Sequential(*[cls(dim=c, num_heads=h, ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                       LayerNorm_type=LayerNorm_type) for _ in range(n)])

        def tblock(c, h):
            return TransformerBlock(dim=c, num_heads=h, ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                    LayerNorm_type=LayerNorm_type)
        F = TransformerResFusionBlock
        self.masa_blk_enc_level1 = blocks(reffusion_n_blocks[0], 2 * dim, heads[0], F)
        self.encoder_level1 = blocks(num_blocks[0], dim, heads[0])
        self.down1_2 = Downsample(dim)
        self.reduce_noise_channel_2 = nn.Conv2d(int(dim * 2 ** 1) + 128, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.masa_blk_enc_level2 = blocks(reffusion_n_blocks[1], 2 * dim * 2 ** 1, heads[1], F)
        self.encoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.reduce_noise_channel_3 = nn.Conv2d(int(dim * 2 ** 2) + 256, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.masa_blk_enc_level3 = blocks(reffusion_n_blocks[2], 2 * dim * 2 ** 2, heads[2], F)
        self.encoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.masa_blk_enc_level4 = blocks(reffusion_n_blocks[3], 2 * dim * 2 ** 3, heads[3], F)
        self.latent = blocks(num_blocks[3], int(dim * 2 ** 3), heads[3])
        self.up4_3 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level3 = nn.Conv2d(int(dim * 2 ** 1) + 192, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.noise_level3 = tblock(int(dim * 2 ** 2) + 512, heads[2])
        self.reduce_noise_level3 = nn.Conv2d(int(dim * 2 ** 2) + 512, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level3 = blocks(num_blocks[2], int(dim * 2 ** 2), heads[2])
        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = nn.Conv2d(int(dim * 2 ** 2), int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.noise_level2 = tblock(int(dim * 2 ** 1) + 224, heads[2])
        self.reduce_noise_level2 = nn.Conv2d(int(dim * 2 ** 1) + 224, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level2 = blocks(num_blocks[1], int(dim * 2 ** 1), heads[1])
        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.noise_level1 = tblock(int(dim * 2 ** 1) + 64, heads[2])
        self.reduce_noise_level1 = nn.Conv2d(int(dim * 2 ** 1) + 64, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.decoder_level1 = blocks(num_blocks[0], int(dim * 2 ** 1), heads[0])
        self.refinement = blocks(num_refinement_blocks, int(dim * 2 ** 1), heads[0])
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim, num_blocks=list(num_blocks),
                        num_refinement_blocks=num_refinement_blocks, heads=list(heads),
                        ffn_expansion_factor=ffn_expansion_factor, bias=bias, LayerNorm_type=LayerNorm_type,
                        decoder=decoder, nf=nf, ext_n_blocks=list(ext_n_blocks),
                        reffusion_n_blocks=list(reffusion_n_blocks), lr_block_size=lr_block_size,
                        ref_down_block_size=ref_down_block_size, dilations=list(dilations), psize=psize)
        for k, p in self.named_parameters():
            if k.startswith(self.unused_parameter_prefixes):
                p.requires_grad_(False)      # the reference never touches them: .grad stays None, the optimiser skips them

    def used_named_parameters(self):
        return [(k, p) for k, p in self.named_parameters() if not k.startswith(self.unused_parameter_prefixes)]

    def check_image_size(self, x):
        mult = self.padder_size * self.lr_block_size
        _, _, h, w = x.shape
        return K.pad_crop(x.contiguous(), -(-h // mult) * mult, -(-w // mult) * mult)

    def forward(self, inp_img, ref_img, noise_emb=None):
        names, params = zip(*self.used_named_parameters())
        return _NetFn.apply(inp_img, ref_img, list(names), self.cfg, *params)

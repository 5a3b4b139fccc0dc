"""Frozen DINOv2 ViT window matcher on the HIP kernels (forward only, no-grad).

Replaces `self.net_ext` of the reference step (models/image_restoration_ref_model.py:74-90, 215-247; network:
models/dino/vision_transformers.py `vit_base(img_size=518, patch_size=14, init_values=1.0, ffn_layer='mlp')`).
Takes the reference's state dict (same key names, strict) and keeps every Linear as a packed 1x1 convolution over
channel-major activations [B, D, 1, T]; LayerNorm is the channel LayerNorm kernel (both use eps 1e-6).
Only the parameter pre-processing (bicubic interpolation of the position embedding to the patch grid of a given
input size, once per size, as vision_transformers.py:179-207 does it) runs in torch on the host.
"""
import math
import os

import torch
import torch.nn.functional as F

from . import kernels as K
from .kernels import PACK_FWD

LN_EPS = 1e-6


# module switches (tests and A/B runs set them; not environment knobs)
LINEAR_MATH = None      # arithmetic of the matcher's Linears when it should differ from kernels.MATH
TOK16 = True            # token-major fp16 pipeline under the fp16 arithmetics (csrc/tdr_tok16.hip)
TOK16X3 = False         # token-major bf16 triple planes at the default arithmetic: built, parity-tested, measured neutral (profiles/r5)
FLAT = True             # batch-flattened token layout

class DinoMatcher:
    def __init__(self, state_dict, device, patch=14, heads=12, interpolate_offset=0.1, linear_math=None):
        """linear_math: arithmetic of the frozen Linears and of the attention (dino.LINEAR_MATH overrides; default: kernels.MATH,
        i.e. the fp32-faithful 2-way fp16 split under TDR_MATH=hx2).  'h1' runs them as ONE fp16 MFMA product per operand pair
        (11-bit operands, fp32 accumulate; the token-major pipeline of csrc/tdr_tok16.hip) instead of the 3 of the split: nothing
        but an arg-max over window similarities leaves this sub-graph, so it is admissible exactly as long as that
        index does not move.  That is pinned on the reference goldens and on a sweep with random-init weights
        (tests/test_hip_dino.py), NOT for trained DINOv2 weights on overlapping windows, where top-1 / top-2 margins can be
        smaller than the ~1e-3 feature error -- hence opt-in (linear_math='h1', 12 - 14 ms per step faster at ref 640^2)."""
        self.linear_math = linear_math or LINEAR_MATH
        if self.linear_math not in (None, 'h1', 'hx2', 'bx3', 'f32'):
            raise ValueError(f'dino.LINEAR_MATH / linear_math: {self.linear_math!r}')
        self.tok16 = False          # set below: the token-major fp16 pipeline (csrc/tdr_tok16.hip), the same 'h1' arithmetic
        sd = {k: v.detach().to(torch.float32) for k, v in state_dict.items()}
        need = ['cls_token', 'pos_embed', 'patch_embed.proj.weight', 'patch_embed.proj.bias', 'norm.weight', 'norm.bias']
        miss = [k for k in need if k not in sd]
        if miss:
            raise KeyError(f'DINOv2 state dict is missing {miss}')
        self.device, self.patch, self.heads, self.offset = device, patch, heads, interpolate_offset
        self.D = sd['cls_token'].shape[-1]
        self.depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
        self.pos_embed = sd['pos_embed']                                 # host copy, interpolated per input size
        self._pos_cache = {}
        dev = lambda t: t.to(device).contiguous()
        self.cls = dev(sd['cls_token'].reshape(-1))
        self.P = {k: dev(v) for k, v in sd.items() if k not in ('pos_embed', 'mask_token')}
        # frozen weights: packed once (the split-bf16 / fp32 layout follows kernels.MATH at construction time)
        self.W = {}
        pw = sd['patch_embed.proj.weight']
        self._pack('patch', pw.reshape(pw.shape[0], -1, 1, 1))
        for i in range(self.depth):
            p = f'blocks.{i}.'
            for name in ('attn.qkv', 'attn.proj', 'mlp.fc1', 'mlp.fc2'):
                w = sd[p + name + '.weight']
                self._pack(p + name, w.reshape(w.shape[0], w.shape[1], 1, 1))
        # 'h1' at ViT-B geometry (head dim 64, widths in GEMM tiles): the blocks run token-major with fp16 operands produced once
        # (dino.TOK16 = False keeps them on the channel-major engines)
        self.tok16 = (self.linear_math == 'h1' and self.D == 64 * heads and self.D % 128 == 0 and self.D <= 1024
                      and TOK16)
        if self.tok16:
            self.W16 = {k: sd[k].to(device).to(torch.float16).contiguous() for k in sd if k.startswith('blocks.') and k.endswith('.weight')
                        and k.split('.')[-2] in ('qkv', 'proj', 'fc1', 'fc2')}
        # Opt-in (dino.TOK16X3 = True): the default arithmetic ('bx3', 6 bf16 products per fp32 product) on the same token-major pipeline.
        # Operands travel as three bf16 planes h | m | l (x = h + m + l exactly), split ONCE by their producers (LayerNorm, GEMM epilogue,
        # the attention's output transpose) instead of once per consuming workgroup; the attention stays on tdr_attention_fwd_math (fp32
        # channel-major q / k / v written by the qkv GEMM's epilogue).  Same numbers as the channel-major engines up to summation order
        # (tests/test_hip_dino.py).  Measured neutral on the matcher-active step (115.9 / 118.0 against 117.7 / 117.3 ms, one box,
        # profiles/r5/bench_dino640_tok16x3_ab.log): the Linears run at 131 - 154 fp32-equivalent TFLOP/s on planes against 125 - 147
        # channel-major (probe_tok16x3_v2.log) -- either layout is bound by the request throughput of its L2 -> LDS operand stream (profiles/r5/pmc_tok16x3.txt:
        # matrix pipe 42 % busy, waves 64 % of their cycles in s_waitcnt) -- and the 12 attention launches (16 ms) are the same kernel in both.
        eff = self.linear_math or K.MATH
        self.tok16x3 = (eff == 'bx3' and K.MATH == 'bx3' and self.D % 128 == 0 and self.D <= 1280
                        and TOK16X3 and TOK16)
        if self.tok16x3:
            self.W3 = {k: K.split_planes3(sd[k].to(device)) for k in sd if k.startswith('blocks.') and k.endswith('.weight')
                       and k.split('.')[-2] in ('qkv', 'proj', 'fc1', 'fc2')}

    def _pack(self, key, w4):
        prev = K.set_pack_plan(None)                      # persistent buffers, not a per-step plan
        try:
            wp, mp, *_ = K.pack_weights(w4.to(self.device).contiguous(), PACK_FWD, math=self.linear_math)
        finally:
            K.set_pack_plan(prev)
        self.W[key] = (wp, mp, w4.shape[0])

    def _pos(self, rows, cols):
        """channel-major [D, 1+T] position embedding for a rows x cols patch grid (vision_transformers.py:179-207)."""
        key = (rows, cols)
        if key not in self._pos_cache:
            pe = self.pos_embed
            n = pe.shape[1] - 1
            if not (rows * cols == n and rows == cols):
                side = int(math.sqrt(n))
                grid = pe[:, 1:].reshape(1, side, side, self.D).permute(0, 3, 1, 2)
                sr, sc = float(rows + self.offset) / math.sqrt(n), float(cols + self.offset) / math.sqrt(n)
                grid = F.interpolate(grid, scale_factor=(sr, sc), mode='bicubic')
                assert grid.shape[-2] == rows and grid.shape[-1] == cols
                pe = torch.cat((pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, self.D)), dim=1)
            self._pos_cache[key] = pe[0].t().contiguous().to(self.device)
        return self._pos_cache[key]

    def _linear(self, x, key, bias, **kw):
        wp, mp, cout = self.W[key]
        return K.conv_forward(x, wp, mp, cout, 1, bias=bias, **kw)

    def tokens(self, x, flat=False):
        """x [B,3,H,W] (multiples of the patch size) -> (final-norm tokens [B, D, LD/32, 32], T): flat column 0 is the
        class token, columns 1..T the patch tokens, the rest padding (kernels.token_ld).
        flat=True computes in the batch-flattened layout [1, D, B*LD/32, 32] (every Linear one GEMM over all B*LD tokens: no
        partly filled pixel tile per image) and returns the same per-image tensor."""
        B, _, H, W = x.shape
        P, D = self.P, self.D
        rows, cols = H // self.patch, W // self.patch
        fb = B if flat else 0
        xp, T = K.patchify(x.contiguous(), self.patch, flat=flat)
        t = K.vit_assemble_(self._linear(xp, 'patch', P['patch_embed.proj.bias']), self.cls, self._pos(rows, cols), T, flat_batch=fb)
        scale = (D // self.heads) ** -0.5
        if self.tok16 and flat:
            return self._blocks_tok16(t, B, T, scale), T
        if self.tok16x3 and flat:
            return self._blocks_tok16x3(t, B, T, scale), T
        for i in range(self.depth):
            p = f'blocks.{i}.'
            h, _, _ = K.layernorm2d_fwd(t, P[p + 'norm1.weight'], P[p + 'norm1.bias'], LN_EPS)
            qkv = self._linear(h, p + 'attn.qkv', P[p + 'attn.qkv.bias'])
            a = K.attention_fwd(qkv, self.heads, scale, T + 1, flat_batch=fb, single_product=self.linear_math == 'h1')
            t = self._linear(a, p + 'attn.proj', P[p + 'attn.proj.bias'], scale=P[p + 'ls1.gamma'], res=t)
            h, _, _ = K.layernorm2d_fwd(t, P[p + 'norm2.weight'], P[p + 'norm2.bias'], LN_EPS)
            h = self._linear(h, p + 'mlp.fc1', P[p + 'mlp.fc1.bias'], relu=2)
            t = self._linear(h, p + 'mlp.fc2', P[p + 'mlp.fc2.bias'], scale=P[p + 'ls2.gamma'], res=t)
        t, _, _ = K.layernorm2d_fwd(t, P['norm.weight'], P['norm.bias'], LN_EPS)
        if flat:                                          # [D][B][LD] -> [B][D][LD] (a layout copy of the final tokens only)
            LD = t.shape[2] * t.shape[3] // B
            t = t.view(D, B, LD).permute(1, 0, 2).contiguous().view(B, D, LD // 32, 32)
        return t, T

    def _blocks_tok16(self, t, B, T, scale):
        """the transformer blocks + final norm on csrc/tdr_tok16.hip: t [1, D, B*LD/32, 32] (channel-major, batch-flattened) ->
        final-norm tokens [B, D, LD/32, 32]"""
        P, D, W16 = self.P, self.D, self.W16
        LD = t.shape[2] * t.shape[3] // B
        x = K.transpose_f32(t.view(1, D, B * LD))[0]                           # residual stream, fp32 [B*LD, D]
        for i in range(self.depth):
            p = f'blocks.{i}.'
            h = K.tok_layernorm(x, P[p + 'norm1.weight'], P[p + 'norm1.bias'], LN_EPS)
            qkv = K.tok16_gemm(h, W16[p + 'attn.qkv.weight'], P[p + 'attn.qkv.bias'])
            a = K.tok16_attention(qkv, B, self.heads, scale, T + 1)
            K.tok16_gemm(a, W16[p + 'attn.proj.weight'], P[p + 'attn.proj.bias'], epi=2, res=x, ls=P[p + 'ls1.gamma'])
            h = K.tok_layernorm(x, P[p + 'norm2.weight'], P[p + 'norm2.bias'], LN_EPS)
            h = K.tok16_gemm(h, W16[p + 'mlp.fc1.weight'], P[p + 'mlp.fc1.bias'], epi=1)
            K.tok16_gemm(h, W16[p + 'mlp.fc2.weight'], P[p + 'mlp.fc2.bias'], epi=2, res=x, ls=P[p + 'ls2.gamma'])
        f = K.tok_layernorm(x, P['norm.weight'], P['norm.bias'], LN_EPS, out_f16=False)
        return K.transpose_f32(f.view(B, LD, D)).view(B, D, LD // 32, 32)

    def _blocks_tok16x3(self, t, B, T, scale):
        """the transformer blocks + final norm on the bf16 triple planes of csrc/tdr_tok16.hip (tdr_tok16x3_gemm): t [1, D, B*LD/32, 32]
        (channel-major, batch-flattened) -> final-norm tokens [B, D, LD/32, 32].  Residual stream fp32 token-major [B*LD, D]."""
        P, D, W3 = self.P, self.D, self.W3
        Pn = t.shape[2] * t.shape[3]
        LD = Pn // B
        x = K.transpose_f32(t.view(1, D, Pn))[0]
        for i in range(self.depth):
            p = f'blocks.{i}.'
            h = K.tok_layernorm(x, P[p + 'norm1.weight'], P[p + 'norm1.bias'], LN_EPS, planes=3)
            qkv = K.tok16x3_gemm(h, W3[p + 'attn.qkv.weight'], P[p + 'attn.qkv.bias'], epi=3)
            a = K.attention_fwd(qkv.view(1, 3 * D, Pn // 32, 32), self.heads, scale, T + 1, flat_batch=B)
            K.tok16x3_gemm(K.cm_to_tok16x3(a.view(D, Pn)), W3[p + 'attn.proj.weight'], P[p + 'attn.proj.bias'], epi=2, out32=x,
                           ls=P[p + 'ls1.gamma'])
            h = K.tok_layernorm(x, P[p + 'norm2.weight'], P[p + 'norm2.bias'], LN_EPS, planes=3)
            h = K.tok16x3_gemm(h, W3[p + 'mlp.fc1.weight'], P[p + 'mlp.fc1.bias'], epi=4, act=2)
            K.tok16x3_gemm(h, W3[p + 'mlp.fc2.weight'], P[p + 'mlp.fc2.bias'], epi=2, out32=x, ls=P[p + 'ls2.gamma'])
        f = K.tok_layernorm(x, P['norm.weight'], P['norm.bias'], LN_EPS, out_f16=False)
        return K.transpose_f32(f.view(B, LD, D)).view(B, D, LD // 32, 32)

    @torch.no_grad()
    def match(self, lq, ref):
        """the lq-sized window of `ref` whose token map is most similar to lq's (:215-247).
        Returns (ref_in [B,C,h,w], index [B] int32, corr [B,N])."""
        B, Cc, h, w = lq.shape
        if h != w:
            raise NotImplementedError('the reference unfolds square (h, h) windows; lq must be square')
        stride = int(h // 4)
        windows, N = K.unfold_windows(ref.contiguous(), h, stride)
        Hd, Wd = int(math.ceil(h / self.patch) * self.patch), int(math.ceil(w / self.patch) * self.patch)
        # one ViT pass over the B images and their B * N windows (the reference runs two, :224-231): every kernel works per image,
        # so the features are the same, and the 4-image pass alone ran its GEMMs / attention at 60 - 75 % of the batched rate
        both = torch.cat([K.resize_bilinear(lq.contiguous(), Hd, Wd), K.resize_bilinear(windows, Hd, Wd)], dim=0)
        f, T = self.tokens(both, flat=FLAT)
        fl, fr = f[:B], f[B:]
        corr, index, ref_in = K.token_match(fl, fr, windows, N, T + 1)
        return ref_in, index, corr


def random_vit_b14_state_dict(seed=0, embed=768, depth=12, grid=37, patch=14):
    """random-init ViT-B/14 state dict with the reference's key names (benchmarks only: no checkpoint is reachable
    offline; the arithmetic does not depend on the weight values)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=0.02: torch.randn(*s, generator=g) * sc
    sd = {'cls_token': r(1, 1, embed), 'pos_embed': r(1, 1 + grid * grid, embed), 'mask_token': torch.zeros(1, embed),
          'patch_embed.proj.weight': r(embed, 3, patch, patch), 'patch_embed.proj.bias': torch.zeros(embed)}
    for i in range(depth):
        p = f'blocks.{i}.'
        sd[p + 'norm1.weight'] = torch.ones(embed); sd[p + 'norm1.bias'] = torch.zeros(embed)
        sd[p + 'attn.qkv.weight'] = r(3 * embed, embed); sd[p + 'attn.qkv.bias'] = torch.zeros(3 * embed)
        sd[p + 'attn.proj.weight'] = r(embed, embed); sd[p + 'attn.proj.bias'] = torch.zeros(embed)
        sd[p + 'ls1.gamma'] = torch.ones(embed)
        sd[p + 'norm2.weight'] = torch.ones(embed); sd[p + 'norm2.bias'] = torch.zeros(embed)
        sd[p + 'mlp.fc1.weight'] = r(4 * embed, embed); sd[p + 'mlp.fc1.bias'] = torch.zeros(4 * embed)
        sd[p + 'mlp.fc2.weight'] = r(embed, 4 * embed); sd[p + 'mlp.fc2.bias'] = torch.zeros(embed)
        sd[p + 'ls2.gamma'] = torch.ones(embed)
    sd['norm.weight'] = torch.ones(embed)
    sd['norm.bias'] = torch.zeros(embed)
    return sd

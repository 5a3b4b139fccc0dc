"""Frozen CLIP ViT image encoder on the HIP kernels (forward only, no-grad) -- SURVEY.md 8a row a28.

Replaces the `CLIPVisionModel` the stage-A trainers call (scripts/train/main_train_i2t_mapping.py:564,726-731;
main_train_tr_mapping.py:609,780; models/image_restoration_text_embed_diffir_model.py:137,264-268): bilinear
resize to the encoder's input size, then `image_features[0]` = the encoder's last hidden state (class token +
patch tokens, no post-layernorm), detached.  The model is third-party (`transformers`); this class takes its state
dict (key names of CLIPVisionModel, with or without the `vision_model.` prefix).

Same layout as dino.py: activations channel-major [B, D, LD/32, 32] (column 0 class token, 1..T patches, rest
padding), every Linear a packed 1x1 convolution (q/k/v projections concatenated into one), LayerNorm the channel
LayerNorm kernel, GELU / quick_gelu a conv epilogue, attention the fused MFMA kernel of csrc/tdr_vit.hip.
"""
import torch

from . import kernels as K
from .kernels import PACK_FWD


def _strip(sd):
    return {(k[len('vision_model.'):] if k.startswith('vision_model.') else k): v.detach().to(torch.float32) for k, v in sd.items()}


TOK16 = False           # module switch: token-major fp16 planes for the frozen CLIP encoder (parity-tested, measured neutral on the stage-A step)

class ClipVisionEncoder:
    def __init__(self, state_dict, device, heads, act='quick_gelu', eps=1e-5):
        """heads / act / eps: CLIPVisionConfig.num_attention_heads / hidden_act / layer_norm_eps
        (ViT-L/14 openai: 16, quick_gelu; ViT-H/14 laion: 16, gelu)."""
        if act not in ('quick_gelu', 'gelu'):
            raise NotImplementedError(f'CLIP hidden_act {act!r}')
        sd = _strip(state_dict)
        need = ['embeddings.class_embedding', 'embeddings.patch_embedding.weight', 'embeddings.position_embedding.weight',
                'pre_layrnorm.weight', 'pre_layrnorm.bias']
        miss = [k for k in need if k not in sd]
        if miss:
            raise KeyError(f'CLIP vision state dict is missing {miss}')
        self.device, self.heads, self.eps = device, heads, eps
        self.act = 3 if act == 'quick_gelu' else 2                  # conv epilogue code (include/tdr.h)
        pw = sd['embeddings.patch_embedding.weight']
        self.D, self.patch = pw.shape[0], pw.shape[-1]
        self.depth = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('encoder.layers.'))
        dev = lambda t: t.to(device).contiguous()
        self.cls = dev(sd['embeddings.class_embedding'].reshape(-1))
        self.pos = dev(sd['embeddings.position_embedding.weight'].t())        # channel-major [D, 1+T]
        self.P = {k: dev(v) for k, v in sd.items() if 'self_attn' not in k and 'embeddings' not in k}
        self.W = {}
        self._pack('patch', pw.reshape(self.D, -1, 1, 1))
        for i in range(self.depth):
            p = f'encoder.layers.{i}.'
            qkv = torch.cat([sd[p + f'self_attn.{n}_proj.weight'] for n in 'qkv'], dim=0)
            self.P[p + 'qkv.bias'] = dev(torch.cat([sd[p + f'self_attn.{n}_proj.bias'] for n in 'qkv'], dim=0))
            self.P[p + 'out.bias'] = dev(sd[p + 'self_attn.out_proj.bias'])
            self._pack(p + 'qkv', qkv.reshape(3 * self.D, self.D, 1, 1))
            w = sd[p + 'self_attn.out_proj.weight']
            self._pack(p + 'out', w.reshape(self.D, self.D, 1, 1))
            for name in ('mlp.fc1', 'mlp.fc2'):
                w = sd[p + name + '.weight']
                self._pack(p + name, w.reshape(w.shape[0], w.shape[1], 1, 1))
        # Opt-in (clip_vision.TOK16 = True): batch-flattened passes under TDR_MATH=hx2 run the blocks token-major on pre-split fp16 planes
        # (csrc/tdr_tok16.hip, tdr_tok16x2_gemm: the same 2-way split arithmetic, operands split once by their producers and fetched
        # as 16-byte fragments).  Measured neutral on the stage-A step (ViT-H 25.5 vs 25.0 ms, ViT-L 20.0 vs 20.5 ms, same box): over
        # ~1 150 token rows both layouts run the Linears at ~140 - 200 fp32-equivalent TFLOP/s, so the channel-major engines with
        # their split-K narrow Linears stay the default.
        inter = sd['encoder.layers.0.mlp.fc1.weight'].shape[0]
        self.tok16 = (K.MATH == 'hx2' and self.D % 128 == 0 and inter % 128 == 0 and self.D <= 1280
                      and TOK16)
        if self.tok16:
            self.W2 = {}
            for i in range(self.depth):
                p = f'encoder.layers.{i}.'
                qkv = torch.cat([sd[p + f'self_attn.{n}_proj.weight'] for n in 'qkv'], dim=0)
                self.W2[p + 'qkv'] = K.split_planes(qkv.to(device))
                self.W2[p + 'out'] = K.split_planes(sd[p + 'self_attn.out_proj.weight'].to(device))
                self.W2[p + 'mlp.fc1'] = K.split_planes(sd[p + 'mlp.fc1.weight'].to(device))
                self.W2[p + 'mlp.fc2'] = K.split_planes(sd[p + 'mlp.fc2.weight'].to(device))

    # Split-K for the narrow-output Linears (attention out-projection, fc2): over the ~1 150 tokens of a 4-image batch they are a
    # single round of ~180 workgroups walking 20 - 80 serial K stages with cold weights (110 / 165 us per launch for ~10 us of
    # work).  Their K chunks run as the "images" of ONE launch (input [1, K, .] viewed as [S, K/S, .], per-chunk packs via wp_ns) and
    # tdr_splitk_finish applies bias / residual / activation to the summed partials (profiles/probe_splitk_1x1.py: 49 -> 33 us and
    # 165 -> 65 us per launch with warm weights).  Batch-flattened layout only (the image axis is what carries the chunks).
    SPLITK = 4

    def _pack(self, key, w4):
        cout, cin = w4.shape[0], w4.shape[1]
        # (the wide Linears -- q/k/v 1280 -> 3840, fc1 -> 5120: 270 - 360 workgroups -- measured flat or slower when split)
        S = self.SPLITK if (self.SPLITK > 1 and cout <= 1536 and cin >= 1024 and cin % (16 * self.SPLITK) == 0) else 1
        prev = K.set_pack_plan(None)                      # persistent buffers, not a per-step plan
        try:
            w4 = w4.to(self.device).contiguous()
            wp, mp, *_ = K.pack_weights(w4, PACK_FWD)
            split = None
            if S > 1:
                kc = cin // S
                packs = [K.pack_weights(w4[:, s * kc:(s + 1) * kc].contiguous(), PACK_FWD)[0] for s in range(S)]
                split = (K.PackedWeights(torch.cat([p.buf for p in packs]), packs[0].fmt), packs[0].buf.numel(), S)
        finally:
            K.set_pack_plan(prev)
        self.W[key] = (wp, mp, cout, split)

    def _linear(self, x, key, bias, res=None, relu=0, **kw):
        wp, mp, cout, split = self.W[key]
        if split is not None and x.shape[0] == 1 and not kw and x.shape[2] * x.shape[3] % 4 == 0:
            wps, per, S = split
            part = K.conv_forward(x.view(S, x.shape[1] // S, x.shape[2], x.shape[3]), wps, mp, cout, 1, wp_ns=per)
            return K.splitk_finish(part, bias=bias, res=res, relu=relu)
        return K.conv_forward(x, wp, mp, cout, 1, bias=bias, res=res, relu=relu, **kw)

    @torch.no_grad()
    def tokens(self, x, flat=False):
        """x [B,3,H,W] at the encoder's input size -> (last hidden state, channel-major [B, D, LD/32, 32], T).
        flat=True: batch-flattened [1, D, B*LD/32, 32] (image b's tokens at columns b*LD ..): every Linear is one GEMM over all
        B*LD tokens -- full pixel tiles and one pass over the weights per layer instead of B."""
        B, _, H, W = x.shape
        P = self.P
        fb = B if flat else 0
        xp, T = K.patchify(x.contiguous(), self.patch, flat=flat)
        if T + 1 != self.pos.shape[1]:
            raise ValueError(f'CLIP encoder built for {self.pos.shape[1] - 1} patches, input gives {T} '
                             '(position embeddings are not interpolated, transformers CLIPVisionEmbeddings)')
        t = K.vit_assemble_(self._linear(xp, 'patch', None), self.cls, self.pos, T, flat_batch=fb)
        scale = (self.D // self.heads) ** -0.5
        if self.tok16 and flat:
            return self._blocks_tok16(t, B, T, scale), T
        t, _, _ = K.layernorm2d_fwd(t, P['pre_layrnorm.weight'], P['pre_layrnorm.bias'], self.eps)
        for i in range(self.depth):
            p = f'encoder.layers.{i}.'
            h, _, _ = K.layernorm2d_fwd(t, P[p + 'layer_norm1.weight'], P[p + 'layer_norm1.bias'], self.eps)
            qkv = self._linear(h, p + 'qkv', P[p + 'qkv.bias'])
            a = K.attention_fwd(qkv, self.heads, scale, T + 1, flat_batch=fb)
            t = self._linear(a, p + 'out', P[p + 'out.bias'], res=t)
            h, _, _ = K.layernorm2d_fwd(t, P[p + 'layer_norm2.weight'], P[p + 'layer_norm2.bias'], self.eps)
            h = self._linear(h, p + 'mlp.fc1', P[p + 'mlp.fc1.bias'], relu=self.act)
            t = self._linear(h, p + 'mlp.fc2', P[p + 'mlp.fc2.bias'], res=t)
        return t, T

    def _blocks_tok16(self, t, B, T, scale):
        """pre-LayerNorm + the encoder layers on csrc/tdr_tok16.hip: t [1, D, B*LD/32, 32] (embeddings, channel-major, batch-flattened)
        -> last hidden state in the same layout.  Residual stream fp32 token-major [B*LD, D]; GEMM operands travel as hi | lo planes;
        the attention stays on tdr_attention_fwd_math (channel-major fp32 q/k/v written by the qkv GEMM's epilogue)."""
        P, D, W2 = self.P, self.D, self.W2
        Pn = t.shape[2] * t.shape[3]
        x = K.tok_layernorm(K.transpose_f32(t.view(1, D, Pn))[0], P['pre_layrnorm.weight'], P['pre_layrnorm.bias'], self.eps, out_f16=False)
        for i in range(self.depth):
            p = f'encoder.layers.{i}.'
            h = K.tok_layernorm(x, P[p + 'layer_norm1.weight'], P[p + 'layer_norm1.bias'], self.eps, planes=True)
            qkv = K.tok16x2_gemm(h, W2[p + 'qkv'], P[p + 'qkv.bias'], epi=3)
            a = K.attention_fwd(qkv.view(1, 3 * D, Pn // 32, 32), self.heads, scale, T + 1, flat_batch=B)
            K.tok16x2_gemm(K.cm_to_tok16x2(a.view(D, Pn)), W2[p + 'out'], P[p + 'out.bias'], epi=2, out32=x)
            h = K.tok_layernorm(x, P[p + 'layer_norm2.weight'], P[p + 'layer_norm2.bias'], self.eps, planes=True)
            h = K.tok16x2_gemm(h, W2[p + 'mlp.fc1'], P[p + 'mlp.fc1.bias'], epi=4, act=self.act)
            K.tok16x2_gemm(h, W2[p + 'mlp.fc2'], P[p + 'mlp.fc2.bias'], epi=2, out32=x)
        return K.transpose_f32(x.view(1, Pn, D)).view(1, D, Pn // 32, 32)

    @torch.no_grad()
    def encode(self, image, size=224, flat=False):
        """the reference call: F.interpolate(image, (224, 224), mode='bilinear') then image_features[0]."""
        x = image.contiguous()
        if tuple(x.shape[-2:]) != (size, size):
            x = K.resize_bilinear(x, size, size)
        return self.tokens(x, flat=flat)


def random_clip_state_dict(hidden=1024, inter=4096, layers=24, patch=14, image=224, seed=0):
    """random-init CLIP ViT state dict with transformers' key names (benchmarks only; ViT-L/14 defaults)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=0.02: torch.randn(*s, generator=g) * sc
    T = (image // patch) ** 2
    sd = {'embeddings.class_embedding': r(hidden), 'embeddings.patch_embedding.weight': r(hidden, 3, patch, patch),
          'embeddings.position_embedding.weight': r(1 + T, hidden),
          'pre_layrnorm.weight': torch.ones(hidden), 'pre_layrnorm.bias': torch.zeros(hidden)}
    for i in range(layers):
        pr = f'encoder.layers.{i}.'
        for nm in ('k_proj', 'v_proj', 'q_proj', 'out_proj'):
            sd[pr + f'self_attn.{nm}.weight'] = r(hidden, hidden); sd[pr + f'self_attn.{nm}.bias'] = torch.zeros(hidden)
        sd[pr + 'layer_norm1.weight'] = torch.ones(hidden); sd[pr + 'layer_norm1.bias'] = torch.zeros(hidden)
        sd[pr + 'mlp.fc1.weight'] = r(inter, hidden); sd[pr + 'mlp.fc1.bias'] = torch.zeros(inter)
        sd[pr + 'mlp.fc2.weight'] = r(hidden, inter); sd[pr + 'mlp.fc2.bias'] = torch.zeros(hidden)
        sd[pr + 'layer_norm2.weight'] = torch.ones(hidden); sd[pr + 'layer_norm2.bias'] = torch.zeros(hidden)
    return sd

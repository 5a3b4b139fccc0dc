"""Stage-A (image-to-text mapping) TRAIN STEP on the HIP kernels -- BASELINE configs[3], SURVEY.md 8d cfg4.

The step of scripts/train/main_train_i2t_mapping.py:704-760, composed from the pieces of rows a28-a30:

    latents = vae.encode(pixel_values) * 0.18215;  noisy = scheduler.add_noise(latents, noise, t)           :706-717
    emb  = image_encoder(interpolate(pixel_values_clip, 224))[0].detach()      CLIP ViT, frozen, no-grad       :726-731
    inj  = mapper([emb])                                                       Mapper, TRAINED                 :733
    ctx  = text_encoder({input_ids, inj_embedding: inj, inj_index})[0]         embedding injection :139-151    :736-738
    pred = unet(noisy, t, {"CONTEXT_TENSOR": ctx}).sample                      injected cross-attention with
                                                                               to_k_global / to_v_global TRAINED :197-233
    loss = mse(pred, noise).mean([1,2,3]).mean(); backward; clip_grad_norm_(mapper.parameters(), 1); AdamW     :744-756

The SD-2.1 VAE / UNet and the CLIP text transformer are third-party `diffusers` / `transformers` models that are absent here
(SURVEY 8c).  SURVEY 8d cfg4 prescribes the substitute: "UNet/VAE replaced by a fixed random linear stub".  `stage_a_stub`
builds that stand-in -- fixed-seed, FROZEN linear maps around the REAL, in-tree parts of the chain:

  * VAE      -> 8x8 average pool + a fixed 3->4 channel map, x 0.18215                       [B,3,512,512] -> [B,4,64,64]
  * text     -> token embedding table + the reference's injection (:139-151) + position embedding (all real), then ONE fixed
                Linear(1024,1024) in place of the 23 transformer layers, then final_layer_norm (real, :176)
  * UNet     -> four levels at the SD cross-attention shapes (tokens 4096 / 1024 / 256 / 64, width 320 / 640 / 1280 / 1280,
                heads 5 / 10 / 20 / 20, head dim 64): a fixed Linear from [f x f pooled latents, 4 time features] to the level
                width, the REAL `inj_forward_crossattention` (frozen to_q / to_out, trainable to_k_global / to_v_global initialised
                as clones of to_k / to_v, :585-593) with its residual, a fixed Linear back to 4 channels, nearest upsampling; the
                level outputs are summed into `pred`.

So every parameter the reference trains (the 40 MLPs of the Mapper and the to_k_global / to_v_global pairs, registered on the
mapper as `{name}_to_k` / `{name}_to_v`) receives its gradient through the same operators as in the reference; only the frozen
context they sit in is reduced.  Host code sequences; all arithmetic is libtdr_hip.so.  No CPU fallback.
"""
import math
import os

import torch
import torch.nn as nn

from . import i2t
from . import kernels as K
from .clip_vision import ClipVisionEncoder
from .kernels import PACK_DGRAD_S1, PACK_FWD
from .optim import FusedClipAdamW
from .parallel import GradAllReducer

# (attention module name with '.' -> '_', latent down-sampling factor, width, heads) -- the four distinct attn2 shapes of SD-2.1
LEVELS = (('down_blocks_0_attentions_0_transformer_blocks_0_attn2', 1, 320, 5),
          ('down_blocks_1_attentions_0_transformer_blocks_0_attn2', 2, 640, 10),
          ('down_blocks_2_attentions_0_transformer_blocks_0_attn2', 4, 1280, 20),
          ('mid_block_attentions_0_transformer_blocks_0_attn2', 8, 1280, 20))
CTX_DIM, SEQ, HEAD_DIM = 1024, 77, 64
VAE_SCALE = 0.18215
LN_EPS = 1e-5


def alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler of the SD-2.1 config ('scaled_linear' betas; diffusers, third party -- published schedule)"""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def stage_a_stub(seed=0, vocab=1024, levels=LEVELS, ctx_dim=CTX_DIM):
    """the frozen stand-in described in the module docstring, as a dict of CPU fp32 tensors (data, not code: the oracle
    consumes the same dict)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    S = {'vae.weight': r(4, 3, sc=1.5), 'alphas_cumprod': alphas_cumprod(),
         'text.token_embedding': r(vocab, ctx_dim, sc=0.5), 'text.position_embedding': r(SEQ, ctx_dim, sc=0.1),
         'text.proj.weight': r(ctx_dim, ctx_dim, sc=ctx_dim ** -0.5), 'text.proj.bias': r(ctx_dim, sc=0.1),
         'text.final_layer_norm.weight': 1 + r(ctx_dim, sc=0.1), 'text.final_layer_norm.bias': r(ctx_dim, sc=0.1)}
    for name, f, dim, heads in levels:
        S[name + '.in.weight'] = r(dim, 8, sc=0.5)
        S[name + '.to_q.weight'] = r(dim, dim, sc=dim ** -0.5)
        S[name + '.to_k.weight'] = r(dim, ctx_dim, sc=ctx_dim ** -0.5)
        S[name + '.to_v.weight'] = r(dim, ctx_dim, sc=ctx_dim ** -0.5)
        S[name + '.to_out.0.weight'] = r(dim, dim, sc=dim ** -0.5)
        S[name + '.to_out.0.bias'] = r(dim, sc=0.1)
        S[name + '.out.weight'] = r(4, dim, sc=dim ** -0.5)
    return S


def synthetic_batch(B, size=512, vocab=1024, num_words=20, seed=0):
    """what UnpairedLQHQDataset hands the step (data/guidance_generation_dataset.py): pixel_values in [-1,1], the CLIP-normalised
    copy, prompt ids with the placeholder position `index`; plus the noise / timesteps the step would draw itself (:709-714)"""
    g = torch.Generator().manual_seed(1234 + seed)
    img = torch.nn.functional.interpolate(torch.rand(B, 3, size // 32, size // 32, generator=g), (size, size), mode='bicubic',
                                          align_corners=False).clamp(0, 1)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    return {'pixel_values': (img * 2 - 1).contiguous(), 'pixel_values_clip': ((img - mean) / std).contiguous(),
            'input_ids': torch.randint(0, vocab, (B, SEQ), generator=g),
            'index': torch.randint(1, SEQ - num_words, (B,), generator=g),
            'noise': torch.randn(B, 4, size // 8, size // 8, generator=g),
            'timesteps': torch.randint(0, 1000, (B,), generator=g)}


def _w4(w):
    return w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous()


# module switches (A/B runs; not environment knobs)
GRAD_SCALE = False            # loss-scaled fp16-split backward of the stage-A step
MAPPER_GROUPED = True          # the Mapper's MLPs as grouped GEMMs

class _Frozen:
    """persistent packs of a frozen Linear: forward (and, on request, the transposed pack of its data gradient)"""

    def __init__(self, w, bias=None, want_dgrad=False):
        self.cout, self.cin = w.shape
        self.bias = bias
        prev = K.set_pack_plan(None)
        try:
            self.fwd = K.pack_weights(_w4(w), PACK_FWD)[:2]
            self.dg = K.pack_weights(_w4(w), PACK_DGRAD_S1)[:2] if want_dgrad else None
        finally:
            K.set_pack_plan(prev)

    def __call__(self, x, **kw):
        return K.conv_forward(x, self.fwd[0], self.fwd[1], self.cout, 1, bias=self.bias, **kw)

    def dgrad(self, dout, **kw):
        return K.conv_forward(dout, self.dg[0], self.dg[1], self.cin, 1, **kw)


class I2TMappingTrainer:
    """`step(batch)` = one iteration of the reference's loop body (:704-760) with accelerate's DDP over the Mapper replaced by the
    RCCL gradient exchange of parallel.GradAllReducer when torch.distributed is initialised (C4, :662).

    clip_state_dict / clip_heads / clip_act: the CLIPVisionModel the reference loads (:564); stub: stage_a_stub(...).
    Optimiser defaults are the script's (:330-360): lr 1e-4 x batch x processes (scale_lr), betas (0.9, 0.999), weight decay
    1e-2, eps 1e-8, constant schedule, clip_grad_norm_ 1.0."""

    def __init__(self, clip_state_dict, clip_heads, stub, clip_act='gelu', num_words=20, lr=1e-4, betas=(0.9, 0.999),
                 weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0, device='cuda', levels=LEVELS, mapper=None, dist_on=None,
                 bucket_mb=64, use_hip_graph=None, clip_image_size=224):
        if not torch.cuda.is_available():
            raise RuntimeError('I2TMappingTrainer: the HIP path needs an MI355X; there is no CPU fallback')
        self.device = torch.device(device)
        self.levels = levels
        self.num_words = num_words
        self.clip_image_size = clip_image_size            # :726 interpolates to (224, 224)
        self.image_encoder = ClipVisionEncoder(clip_state_dict, self.device, clip_heads, act=clip_act)
        dev = lambda t: t.to(self.device, torch.float32).contiguous()
        S = self.S = {k: dev(v) for k, v in stub.items()}
        self.mapper = (mapper if mapper is not None else i2t.Mapper(self.image_encoder.D, CTX_DIM, num_words)).to(self.device)
        # :571-593 -- to_k_global / to_v_global start as clones of the frozen to_k / to_v and live on the mapper
        for name, _, dim, _ in levels:
            for kv in ('k', 'v'):
                if not hasattr(self.mapper, f'{name}_to_{kv}'):
                    lin = nn.Linear(CTX_DIM, dim, bias=False)
                    lin.weight.data = S[f'{name}.to_{kv}.weight'].clone()
                    self.mapper.add_module(f'{name}_to_{kv}', lin.to(self.device))
        self.train_kv = True
        self._trainables()
        self.mlp_names = [k for k in self.names if k.startswith('mapping_')]
        self.optimizer = FusedClipAdamW([{'params': self.params}], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                        max_norm=max_grad_norm, use_grad_clip=True)
        # dist_on: average the gradients over the ranks of the initialised process group (accelerate's DDP over the Mapper, :662).
        # None (default): exactly when a process group of more than one rank exists -- as accelerate does; True without a process
        # group is an error; False keeps this trainer local even under a launcher, and says so once (unsynchronised replicas are
        # almost never what a torchrun launch wants).  Decided BEFORE the reducer exists: a local trainer never brings RCCL up.
        pg_up = torch.distributed.is_available() and torch.distributed.is_initialized()
        pg_world = torch.distributed.get_world_size() if pg_up else 1
        if dist_on is None:
            dist_on = pg_world > 1
        self.dist = bool(dist_on)
        if self.dist and not pg_up:
            raise RuntimeError('I2TMappingTrainer(dist_on=True): torch.distributed is not initialised')
        if not self.dist and pg_world > 1:
            import warnings
            warnings.warn(f'I2TMappingTrainer(dist_on=False) inside a process group of {pg_world} ranks: gradients are NOT averaged, '
                          'every rank trains its own replica', RuntimeWarning, stacklevel=2)
        self.reducer = GradAllReducer(list(zip(self.names, self.params)), bucket_mb=bucket_mb,
                                      local_only=not self.dist and os.environ.get('TDR_FORCE_COLLECTIVES') != '1')
        self._plan = K.PackPlan()
        # the Mapper as G-way grouped GEMMs over batch-flattened tokens (i2t.mapper_fwd_grouped); stage_a.MAPPER_GROUPED = False keeps the
        # 2 x num_words chains of small launches on four stream lanes
        self.grad_scale = GRAD_SCALE
        self.grouped = MAPPER_GROUPED
        self.stacks = i2t.MapperStacks(self.mapper) if self.grouped else None
        # frozen stand-ins, packed once
        self.vae = _Frozen(S['vae.weight'] * (VAE_SCALE / 64.0))              # average pool = block sum / 64, folded in
        self.text_proj = _Frozen(S['text.proj.weight'], S['text.proj.bias'], want_dgrad=True)
        self.lv = {}
        for name, f, dim, heads in levels:
            self.lv[name] = dict(inp=_Frozen(S[name + '.in.weight']), q=_Frozen(S[name + '.to_q.weight']),
                                 o=_Frozen(S[name + '.to_out.0.weight'], S[name + '.to_out.0.bias'], want_dgrad=True),
                                 out=_Frozen(S[name + '.out.weight'], want_dgrad=True))
        self.use_hip_graph = (os.environ.get('TDR_GRAPH', '1') == '1') if use_hip_graph is None else bool(use_hip_graph)
        self._g = None
        self._eager_left = 2

    def _trainables(self):
        """the tensors the optimiser owns (registration order): here every parameter of the Mapper, incl. the to_k / to_v it carries"""
        self.names = [k for k, _ in self.mapper.named_parameters()]
        self.params = [p for _, p in self.mapper.named_parameters()]

    def _words_fwd(self, tok, T, B, P):
        if self.grouped:
            return i2t.mapper_fwd_grouped(tok, B, T, self.stacks)
        return i2t.mapper_fwd(tok, T, P, self.num_words)

    def _words_bwd(self, dinj, saved, P):
        return i2t.mapper_bwd_grouped(dinj, self.stacks, saved) if self.grouped else i2t.mapper_bwd(dinj, P, self.num_words, saved)

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def _lin(x, w, **kw):
        wp, mp, *_ = K.pack_weights(_w4(w), PACK_FWD)
        return K.conv_forward(x, wp, mp, w.shape[0], 1, **kw)

    @staticmethod
    def _dgrad(dout, w, **kw):
        wp, mp, *_ = K.pack_weights(_w4(w), PACK_DGRAD_S1)
        return K.conv_forward(dout, wp, mp, w.shape[1], 1, **kw)

    def _fwd_bwd(self, b):
        """forward, MSE, hand-written backward; parameter gradients land in the reducer's arena.  Returns loss [1]."""
        S, P = self.S, {k: p.data for k, p in self.mapper.named_parameters()}
        prev_plan = K.set_pack_plan(self._plan)
        prev_scaled = K.GRAD_SCALED
        try:
            # Optional loss-scaled backward (stage_a.GRAD_SCALE = True; off by default): under TDR_MATH=hx2 the gradient GEMMs may take the
            # 2-way fp16 split (3 products instead of the 6 of the 3-way bf16 split) if their operands sit in the fp16 window.
            # Unlike the restoration step this backward spans ~2^20: dpred = 2 (pred - noise) / numel ~ 2^-14 while dk / dv sum over
            # 4096 queries and reach ~2^5, so the exact power-of-two scale is S = numel / 256 (2^8 at bs 4): the largest operands
            # stay below 2^14, the smallest lose part of their residual plane.  Every backward kernel is linear in the gradient, the
            # gather into the arena multiplies by 1 / S, and the optimiser's device-resident guard skips a step whose gradient norm
            # is not finite (and halves S).  Measured 29.7 -> 27.9 ms; no range survey guards it here, hence opt-in.
            numel = b['noise'].numel()
            gs = 2.0 ** (math.floor(math.log2(numel)) - 8) if (K.fp16_path() and self.grad_scale) else 1.0
            K.set_grad_scaled(gs != 1.0)
            guard = self.optimizer.ensure_guard(self.device)
            if not torch.cuda.is_current_stream_capturing():
                guard.set_max_scale(gs)
            self.reducer.guard = guard
            self._plan.run()
            t, idx, ids = b['timesteps'], b['index'], b['input_ids']
            # ---- frozen front: VAE stand-in, forward diffusion, CLIP image encoder (no-grad)
            lat = self.vae(K.pool_sum(b['pixel_values'], 8))
            noisy = K.add_noise(lat, b['noise'], t, S['alphas_cumprod'])
            tok, T = self.image_encoder.encode(b['pixel_values_clip'], size=self.clip_image_size, flat=self.grouped)
            # ---- Mapper (a29) and the text side: injection (:139-151), stand-in projection, final_layer_norm
            inj, msaved = self._words_fwd(tok, T, b['pixel_values_clip'].shape[0], P)
            new = K.text_inject_fwd(ids, S['text.token_embedding'], S['text.position_embedding'], inj, idx)
            z = self.text_proj(new)
            ctx, mu, rs = K.layernorm2d_fwd(z, S['text.final_layer_norm.weight'], S['text.final_layer_norm.bias'], LN_EPS)
            # ---- UNet stand-in: the injected cross-attention (a30) at the four SD shapes
            pred = torch.empty_like(b['noise'])
            saved = []
            for li, (name, f, dim, heads) in enumerate(self.levels):
                L = self.lv[name]
                h = L['inp'](K.pool_time(noisy, t, f))                                         # [B, dim, s, s] = channel-major tokens
                q = L['q'](h)
                wk, wv = P[f'{name}_to_k.weight'], P[f'{name}_to_v.weight']
                k, v = self._lin(ctx, wk), self._lin(ctx, wv)
                Tq = h.shape[2] * h.shape[3]
                a, lse = K.cross_attention_fwd(q, k, v, heads, HEAD_DIM ** -0.5, Tq, SEQ)
                o = L['o'](a, res=h)                                                          # h + to_out(attention)
                K.upsample_nearest_add_(pred, L['out'](o), f, accumulate=li > 0)
                saved.append((q, k, v, a, lse, Tq))
            loss, dpred = K.pixel_loss(K.LOSS_MSE, pred, b['noise'], 1.0, 0.0, guard=guard)
            # ---- backward: only what leads to a trained parameter (the UNet side is frozen: no dq, no dh)
            sink = self.reducer.begin(defer_collectives=True)
            G = {}
            dctx = None
            for (name, f, dim, heads), (q, k, v, a, lse, Tq) in zip(self.levels, saved):
                L = self.lv[name]
                do = L['out'].dgrad(K.pool_sum(dpred, f))
                da = L['o'].dgrad(do)
                _, dk, dv = K.cross_attention_bwd(q, k, v, a, da, lse, heads, HEAD_DIM ** -0.5, Tq, SEQ, need_dq=False)
                wk, wv = P[f'{name}_to_k.weight'], P[f'{name}_to_v.weight']
                if self.train_kv:
                    G[f'{name}_to_k.weight'] = K.conv_wgrad(ctx, dk, dim, CTX_DIM, 1).view(dim, CTX_DIM)
                    G[f'{name}_to_v.weight'] = K.conv_wgrad(ctx, dv, dim, CTX_DIM, 1).view(dim, CTX_DIM)
                dctx = self._dgrad(dk, wk, res=dctx) if dctx is not None else self._dgrad(dk, wk)
                dctx = self._dgrad(dv, wv, res=dctx)
            dz, _, _ = K.layernorm2d_bwd(dctx, z, mu, rs, S['text.final_layer_norm.weight'])
            dinj = K.text_inject_bwd(self.text_proj.dgrad(dz), idx, SEQ, self.num_words)
            G.update(self._words_bwd(dinj, msaved, P))
            for kname in self.names:                      # fixed arrival order = registration order
                sink[kname] = G[kname]
            grads = self.reducer.finish()
        finally:
            K.set_grad_scaled(prev_scaled)
            K.set_pack_plan(prev_plan)
            self._plan.invalidate()                       # the optimiser is about to change the weights
        if not getattr(self, '_bound', False) or self.reducer.relaid:
            for kname, p in zip(self.names, self.params):
                p.grad = grads[kname]
            self._bound = True
        self.pred = pred
        return loss

    # ------------------------------------------------------------------ step
    def _device_batch(self, batch):
        d = self.device
        out = {k: batch[k].to(d, torch.float32).contiguous() for k in ('pixel_values', 'pixel_values_clip', 'noise')}
        for k in ('input_ids', 'index', 'timesteps'):
            out[k] = batch[k].to(d, torch.int32).contiguous()
        return out

    def _eager(self, b):
        loss = self._fwd_bwd(b)
        self.reducer.allreduce_flat()
        self.optimizer.step()
        return loss

    def step(self, batch):
        """batch: pixel_values, pixel_values_clip [B,3,H,W], input_ids [B,77], index [B], and -- drawn here when absent, as the
        reference draws them (:709-714) -- noise [B,4,H/8,W/8], timesteps [B].  Returns the loss tensor [1] (device)."""
        if 'noise' not in batch:
            B, _, H, W = batch['pixel_values'].shape
            batch = dict(batch, noise=torch.randn(B, 4, H // 8, W // 8, device=self.device),
                         timesteps=torch.randint(0, 1000, (B,), device=self.device))
        b = self._device_batch(batch)
        if not self.use_hip_graph:
            return self._eager(b)
        key = tuple(b['pixel_values'].shape)
        g = self._g
        if g is None or g['key'] != key:
            if self._eager_left > 0:                      # allocator / workspace / arena-layout warm-up
                self._eager_left -= 1
                return self._eager(b)
            g = self._g = {'key': key, 'in': {k: v.clone() for k, v in b.items()}, 'refs': []}
            # what torch.cuda.graph() does on entry: hand the eager steps' cached blocks back, or the graphs' private pool has to
            # fit NEXT to them (PromptIR-ref 384x384 bs 8: 106 GB live + 180 GB cached = out of memory)
            torch.cuda.synchronize()
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            pool = torch.cuda.graph_pool_handle()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap), K.workspace_capture(g['refs']):
                gA = torch.cuda.CUDAGraph()
                gA.capture_begin(pool=pool, capture_error_mode='thread_local')
                try:
                    g['loss'] = self._fwd_bwd(g['in'])
                finally:
                    gA.capture_end()
                g['pinned'] = self.reducer.pinned_tables
                self.optimizer.prepare()
                gB = torch.cuda.CUDAGraph()
                gB.capture_begin(pool=pool, capture_error_mode='thread_local')
                try:
                    self.optimizer.launch()
                finally:
                    gB.capture_end()
            torch.cuda.current_stream().wait_stream(cap)
            g['A'], g['B'] = gA, gB
        else:
            for k, v in b.items():
                g['in'][k].copy_(v, non_blocking=True)
            self.optimizer.prepare()
        g['A'].replay()
        self.reducer.allreduce_flat()
        g['B'].replay()
        return g['loss']


class TRMappingTrainer(I2TMappingTrainer):
    """The textual-restoration mapping step (scripts/train/main_train_tr_mapping.py:757-812): the same pipeline as the
    image-to-text step with the Mapper -- and the `to_k_global` / `to_v_global` it carries -- FROZEN (:668) and a `CleanMapper`
    (:84-120, i2t.CleanMapper) between the Mapper's words and the placeholder-token injection (:785-786); only the CleanMapper
    requires gradients (:671).

    Reference defect R9 (recorded in DESIGN.md): the script builds its optimiser over `mapper.parameters()` (:679-685) -- all
    frozen -- and clips / zeroes the same list (:806-810), so as written a step changes no parameter and the CleanMapper's
    gradients accumulate.  `as_written=True` restates exactly that (gradients accumulate into `clean_mapper.*.grad`, no update);
    the default is the evident intent: AdamW (+ clip_grad_norm_ 1.0, zero_grad) over the CleanMapper."""

    def __init__(self, clip_state_dict, clip_heads, stub, clean_mapper=None, as_written=False, **kw):
        self._clean_arg, self.as_written = clean_mapper, bool(as_written)
        super().__init__(clip_state_dict, clip_heads, stub, **kw)

    def _trainables(self):
        cm = self._clean_arg if self._clean_arg is not None else i2t.CleanMapper(CTX_DIM, CTX_DIM, self.num_words)
        self.clean_mapper = cm.to(self.device)
        for p in self.mapper.parameters():
            p.requires_grad_(False)                                   # :668 freeze_params(mapper.parameters())
        self.train_kv = False
        self.names = [k for k, _ in self.clean_mapper.named_parameters()]
        self.params = [p for _, p in self.clean_mapper.named_parameters()]
        self._accum = None

    def _words_fwd(self, tok, T, B, P):
        if not self.grouped:
            raise NotImplementedError('TRMappingTrainer runs the grouped kernels (stage_a.MAPPER_GROUPED)')
        inj, _ = i2t.mapper_fwd_grouped(tok, B, T, self.stacks)      # frozen: nothing kept for a backward pass
        return i2t.clean_mapper_fwd_grouped(inj, self.clean_mapper.stacks())

    def _words_bwd(self, dinj, saved, P):
        G, _ = i2t.clean_mapper_bwd_grouped(dinj, self.clean_mapper.stacks(), saved)
        return G

    def _eager(self, b):
        loss = self._fwd_bwd(b)
        self.reducer.allreduce_flat()
        if self.as_written:
            self._accumulate()
        else:
            self.optimizer.step()
        return loss

    def _accumulate(self):
        """as written: nothing zeroes the CleanMapper's gradients, nothing updates it"""
        if self._accum is None:
            self._accum = {k: p.grad.clone() for k, p in zip(self.names, self.params)}
        else:
            for k, p in zip(self.names, self.params):
                self._accum[k] += p.grad

    def step(self, batch):
        if self.as_written:
            if 'noise' not in batch:
                B, _, H, W = batch['pixel_values'].shape
                batch = dict(batch, noise=torch.randn(B, 4, H // 8, W // 8, device=self.device),
                             timesteps=torch.randint(0, 1000, (B,), device=self.device))
            return self._eager(self._device_batch(batch))
        return super().step(batch)

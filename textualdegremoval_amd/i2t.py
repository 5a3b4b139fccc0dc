"""Stage-A image-to-text mapping pieces on the HIP kernels -- SURVEY.md 8a rows a29 / a30.

`Mapper` mirrors the class the reference defines in scripts/train/main_train_i2t_mapping.py:40-81 (also
main_train_tr_mapping.py): same constructor, same parameter names (`mapping_{i}.{0,1,3,4,6,7,9}.*`,
`mapping_patch_{i}.*`), same default init (the nn.Linear / nn.LayerNorm members are parameter containers; their ATen
forward is never called).  Forward and backward run in libtdr_hip.so: every Linear is a 1x1 convolution over
channel-major tokens, nn.LayerNorm the channel LayerNorm kernel (eps 1e-5), LeakyReLU / class-token gather /
token-mean combine the kernels of csrc/tdr_i2t.hip.

`CrossAttentionFn` / `cross_attention` restate `inj_forward_crossattention` (:197-233).
"""
import torch
import torch.nn as nn

from . import kernels as K
from .kernels import PACK_DGRAD_S1, PACK_FWD

LN_EPS = 1e-5          # nn.LayerNorm default
SLOPE = 0.01           # nn.LeakyReLU default


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: the HIP path needs tensors on the MI355X; there is no CPU fallback')


def _lin_fwd(x, w, b):
    wp, mp, *_ = K.pack_weights(w.view(w.shape[0], w.shape[1], 1, 1), PACK_FWD)
    return K.conv_forward(x, wp, mp, w.shape[0], 1, bias=b)


def _lin_bwd(dout, x, w, need_dx):
    """returns (dx or None, dw [out,in], db [out])."""
    Cout, Cin = w.shape
    gw, gb = K.conv_wgrad(x, dout, Cout, Cin, 1, want_db=True)
    dx = None
    if need_dx:
        wp, mp, *_ = K.pack_weights(w.view(Cout, Cin, 1, 1), PACK_DGRAD_S1)
        dx = K.conv_forward(dout, wp, mp, Cin, 1)
    return dx, gw.view(Cout, Cin), gb


def mlp_fwd(x, P, pre):
    """Linear-LayerNorm-LeakyReLU x3 + Linear (:53-62) on channel-major tokens x [n, Din, h, 32]."""
    saved = []
    for j in (0, 3, 6):
        z = _lin_fwd(x, P[f'{pre}{j}.weight'], P[f'{pre}{j}.bias'])
        zn, mu, rs = K.layernorm2d_fwd(z, P[f'{pre}{j + 1}.weight'], P[f'{pre}{j + 1}.bias'], LN_EPS)
        y = K.leaky_relu_fwd(zn, SLOPE)
        saved.append((x, z, mu, rs, y))
        x = y
    return _lin_fwd(x, P[f'{pre}9.weight'], P[f'{pre}9.bias']), (saved, x)


def mlp_bwd(dout, P, pre, saved, G):
    """parameter gradients into G (the input embedding is detached in the reference, :731)."""
    layers, x_last = saved
    d, G[f'{pre}9.weight'], G[f'{pre}9.bias'] = _lin_bwd(dout, x_last, P[f'{pre}9.weight'], True)
    for j, (x, z, mu, rs, y) in zip((6, 3, 0), reversed(layers)):
        dzn = K.leaky_relu_bwd(d, y, SLOPE)
        dz, G[f'{pre}{j + 1}.weight'], G[f'{pre}{j + 1}.bias'] = K.layernorm2d_bwd(dzn, z, mu, rs, P[f'{pre}{j + 1}.weight'])
        d, G[f'{pre}{j}.weight'], G[f'{pre}{j}.bias'] = _lin_bwd(dz, x, P[f'{pre}{j}.weight'], j > 0)


LANES = 4   # HIP streams the 2 x num_words independent MLPs are spread over


def mapper_fwd(tok, T, P, num_words):
    """tok [B, Din, LD/32, 32] channel-major (column 0 class token, 1..T patches) -> ([B, words, Dout], saved).
    The words are independent chains of small launches (a 1280 x 1280 Linear over 4 x 257 tokens fills a third of the chip, the
    class-token MLPs are pure latency): word i runs on lane i % LANES (kernels.lane), forward and backward on the same lane."""
    B = tok.shape[0]
    if B > 32:
        raise NotImplementedError('HIP Mapper: batch <= 32 per call (class tokens travel as one 32-pixel row)')
    cls_in = K.gather_col(tok, 0)                                     # embs[:, :1]
    dout_dim = P['mapping_0.9.weight'].shape[0]
    out = torch.empty(B, num_words, dout_dim, dtype=torch.float32, device=tok.device)
    saved = []
    for i in range(num_words):
        with K.lane(i % LANES):
            c, sv_c = mlp_fwd(cls_in, P, f'mapping_{i}.')
            p, sv_p = mlp_fwd(tok, P, f'mapping_patch_{i}.')
            K.mapper_combine(c, p, T, out, i)
            saved.append((sv_c, sv_p, c, p))                          # (c, p stay referenced: their lane may still be reading them)
    K.lanes_join()
    return out, (saved, tok.shape[2] * tok.shape[3], T)


def mapper_bwd(dout, P, num_words, saved):
    sv, LD, T = saved
    G = {}
    dout = dout.contiguous()
    cur = torch.cuda.current_stream()
    for i in range(num_words):
        with K.lane(i % LANES):
            dc, dp = K.mapper_combine_bwd(dout, LD, T, i)
            Gi = {}
            mlp_bwd(dc, P, f'mapping_{i}.', sv[i][0], Gi)
            mlp_bwd(dp, P, f'mapping_patch_{i}.', sv[i][1], Gi)
            for g in Gi.values():
                g.record_stream(cur)                                  # consumed by the caller's stream after the join
            G.update(Gi)
    K.lanes_join()
    return G


# ---------------------------------------------------------------------------- the same Mapper as G-way grouped GEMMs
# 40 chains of small launches (~1 200 per forward + backward) leave the chip mostly idle: a 1280 x 1280 Linear over 4 x 257 tokens
# is a third of a round of workgroups, a class-token MLP is pure weight-streaming latency.  Grouped, every layer of the 20 patch MLPs
# (and of the 20 class-token MLPs) is ONE launch: tdr_conv_forward / tdr_conv_wgrad with N = words "images", per-image packed
# weights (wp_ns) and biases, the shared first-layer input addressed with image stride 0, LayerNorm + LeakyReLU fused and per-word
# (tdr_group_ln_act_*).  Tokens are batch-flattened: [words, C, B*LD/32, 32].
KINDS = ('mapping_', 'mapping_patch_')


class MapperStacks:
    """Re-points every parameter of a Mapper into per-layer stacks [words, ...] (p.data becomes a view of the stack, so the
    optimiser's in-place updates, load_state_dict and state_dict keep working on the same storage) -- the grouped kernels then
    address word g of a layer as base + g * stride."""

    def __init__(self, mapper, kinds=KINDS):
        G = self.G = mapper.num_words
        self._slots = []
        P = dict(mapper.named_parameters())
        self.W, self.B, self.LW, self.LB = {}, {}, {}, {}
        for kind in kinds:
            for j in (0, 3, 6, 9):
                w0 = P[f'{kind}0.{j}.weight']
                self.W[kind, j] = torch.empty(G, *w0.shape, dtype=torch.float32, device=w0.device)
                self.B[kind, j] = torch.empty(G, w0.shape[0], dtype=torch.float32, device=w0.device)
            for j in (1, 4, 7):
                c = P[f'{kind}0.{j}.weight'].shape[0]
                self.LW[kind, j] = torch.empty(G, c, dtype=torch.float32, device=w0.device)
                self.LB[kind, j] = torch.empty(G, c, dtype=torch.float32, device=w0.device)
            for i in range(G):
                for j in (0, 3, 6, 9):
                    self._adopt(P[f'{kind}{i}.{j}.weight'], self.W[kind, j][i])
                    self._adopt(P[f'{kind}{i}.{j}.bias'], self.B[kind, j][i])
                for j in (1, 4, 7):
                    self._adopt(P[f'{kind}{i}.{j}.weight'], self.LW[kind, j][i])
                    self._adopt(P[f'{kind}{i}.{j}.bias'], self.LB[kind, j][i])

    def _adopt(self, p, slot):
        with torch.no_grad():
            slot.copy_(p.data)
            p.data = slot
        self._slots.append((p, slot))

    def verify(self):
        """every parameter must still live in its stack slot: mapper.to() / .float() / .cuda() or a wrapper that re-allocates
        parameters detaches them, and the grouped kernels would go on reading (and the optimiser stop updating) orphaned
        buffers without any error.  Cheap (host-side pointer compares); called before every grouped forward."""
        for p, slot in self._slots:
            if p.data_ptr() != slot.data_ptr():
                raise RuntimeError('MapperStacks: a Mapper parameter no longer aliases its stack slot (the module was moved or its '
                                   'parameters were re-allocated after the stacks were built): rebuild with MapperStacks(mapper)')


def _chain_fwd(x, st, kind):
    """x [G (stride 0), Din, H, W] -> (out [G, Dout, H, W], saved)"""
    saved = []
    for j in (0, 3, 6):
        wp, mp, per = K.pack_weights_grouped(st.W[kind, j], PACK_FWD)
        z = K.conv_forward(x, wp, mp, st.W[kind, j].shape[1], 1, wp_ns=per, bias=st.B[kind, j])
        y, mu, rs = K.group_ln_act_fwd(z, st.LW[kind, j + 1], st.LB[kind, j + 1], LN_EPS, SLOPE)
        saved.append((x, z, mu, rs, y))
        x = y
    wp, mp, per = K.pack_weights_grouped(st.W[kind, 9], PACK_FWD)
    return K.conv_forward(x, wp, mp, st.W[kind, 9].shape[1], 1, wp_ns=per, bias=st.B[kind, 9]), (saved, x)


def _lin_bwd_grouped(dout, x, W, need_dx):
    """-> (dx or None, dW [G, out, in]); the bias gradient (pixel sums of dout) comes from dout's producer"""
    G, Cout, Cin = W.shape
    gw = K.conv_wgrad(x, dout, Cout, Cin, 1, per_image=True).view(G, Cout, Cin)
    dx = None
    if need_dx:
        wp, mp, per = K.pack_weights_grouped(W, PACK_DGRAD_S1)
        dx = K.conv_forward(dout, wp, mp, Cin, 1, wp_ns=per)
    return dx, gw


def _chain_bwd(d, dsum, st, kind, saved, out, need_dx=False):
    """d: gradient of the chain's output [G, Dout, H, W], dsum [G, Dout] its pixel sums; returns the gradient of the chain's
    input when `need_dx` (CleanMapper: the Mapper's words), else None"""
    layers, x_last = saved
    out[kind, 9, 'bias'] = dsum
    d, out[kind, 9, 'weight'] = _lin_bwd_grouped(d, x_last, st.W[kind, 9], True)
    for j, (x, z, mu, rs, y) in zip((6, 3, 0), reversed(layers)):
        dz, out[kind, j + 1, 'weight'], out[kind, j + 1, 'bias'], out[kind, j, 'bias'] = \
            K.group_ln_act_bwd(d, y, z, mu, rs, st.LW[kind, j + 1], SLOPE)
        d, out[kind, j, 'weight'] = _lin_bwd_grouped(dz, x, st.W[kind, j], j > 0 or need_dx)
    return d if need_dx else None


def mapper_fwd_grouped(tok, B, T, st):
    """tok: batch-flattened channel-major tokens [1, Din, B*LD/32, 32] (ClipVisionEncoder.encode(..., flat=True));
    -> ([B, words, Dout], saved)"""
    if B > 32:
        raise NotImplementedError('HIP Mapper: batch <= 32 per call (class tokens travel as one 32-pixel row)')
    st.verify()
    G = st.G
    LD = tok.shape[2] * tok.shape[3] // B
    cls_in = K.gather_col_flat(tok, B, 0)                                            # embs[:, :1]
    c, sv_c = _chain_fwd(cls_in.expand(G, -1, -1, -1), st, 'mapping_')
    p, sv_p = _chain_fwd(tok.expand(G, -1, -1, -1), st, 'mapping_patch_')
    return K.mapper_combine_all(c, p, B, LD, T), (sv_c, sv_p, B, LD, T)


def mapper_bwd_grouped(dout, st, saved):
    """-> {parameter name: gradient} for every MLP parameter (views of per-layer [words, ...] gradient stacks)"""
    sv_c, sv_p, B, LD, T = saved
    dc, dp, dsum = K.mapper_combine_all_bwd(dout.contiguous(), LD, T)
    stacks = {}
    _chain_bwd(dc, dsum, st, 'mapping_', sv_c, stacks)
    _chain_bwd(dp, dsum, st, 'mapping_patch_', sv_p, stacks)
    G = {}
    for (kind, j, what), g in stacks.items():
        for i in range(st.G):
            G[f'{kind}{i}.{j}.{what}'] = g[i]
    return G


# ---------------------------------------------------------------------------- CleanMapper (main_train_tr_mapping.py:84-120)
def _words_to_cm(x):
    """[B, words, D] token-major -> [words, D, 1, 32] channel-major: word g is "image" g, sample b its pixel b (zero beyond B)"""
    B, G, D = x.shape
    return K.transpose_pad(x.contiguous().view(1, B, G * D), 32).view(G, D, 1, 32)


def _cm_to_words(x, B):
    """[words, D, 1, 32] -> [B, words, D]"""
    G, D = x.shape[0], x.shape[1]
    return K.transpose_pad(x.reshape(1, G * D, 32), G * D)[0, :B].reshape(B, G, D)


def clean_mapper_fwd_grouped(inj, st):
    """inj [B, words, Din] (the Mapper's words) -> ([B, words, Dout], saved): word i through its own MLP `mapping_{i}` -- the
    class-token chain of the grouped Mapper with a per-word input (every layer ONE grouped launch over the words)."""
    B = inj.shape[0]
    if B > 32:
        raise NotImplementedError('HIP CleanMapper: batch <= 32 per call (the samples of a word travel as one 32-pixel row)')
    st.verify()
    out, sv = _chain_fwd(_words_to_cm(inj), st, 'mapping_')
    return _cm_to_words(out, B), (sv, B)


def clean_mapper_bwd_grouped(dout, st, saved, need_dinj=False):
    """-> ({parameter name: gradient}, d inj [B, words, Din] or None)"""
    sv, B = saved
    d = _words_to_cm(dout)
    G, Dout = d.shape[0], d.shape[1]
    dsum = K.channel_sum(d.view(1, G * Dout, 1, 32)).view(G, Dout)          # bias gradient of the last Linear (padded pixels are 0)
    stacks = {}
    dinj = _chain_bwd(d, dsum, st, 'mapping_', sv, stacks, need_dx=need_dinj)
    Gd = {}
    for (kind, j, what), g in stacks.items():
        for i in range(st.G):
            Gd[f'{kind}{i}.{j}.{what}'] = g[i]
    return Gd, (_cm_to_words(dinj, B) if need_dinj else None)


class _CleanMapperFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inj, st, names, *params):
        _require_gpu(inj, 'CleanMapper')
        out, saved = clean_mapper_fwd_grouped(inj.detach().contiguous(), st)
        ctx.st, ctx.saved, ctx.names, ctx.need = st, saved, names, inj.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        G, dinj = clean_mapper_bwd_grouped(dout.contiguous(), ctx.st, ctx.saved, need_dinj=ctx.need)
        ctx.saved = None
        return (dinj, None, None) + tuple(G[k] for k in ctx.names)


class CleanMapper(nn.Module):
    """`CleanMapper` of scripts/train/main_train_tr_mapping.py:84-120 (same constructor, parameter names and default init): word i
    of the injected embedding through `mapping_{i}`.  Forward / backward on the grouped kernels (the parameters are re-pointed
    into per-layer stacks on first use: MapperStacks)."""

    def __init__(self, input_dim: int, output_dim: int, num_words: int):
        super().__init__()
        self.num_words = num_words
        for i in range(self.num_words):
            setattr(self, f'mapping_{i}', nn.Sequential(nn.Linear(input_dim, 1280), nn.LayerNorm(1280), nn.LeakyReLU(),
                                                        nn.Linear(1280, 1280), nn.LayerNorm(1280), nn.LeakyReLU(),
                                                        nn.Linear(1280, 1280), nn.LayerNorm(1280), nn.LeakyReLU(),
                                                        nn.Linear(1280, output_dim)))
        self._stacks = None

    def stacks(self):
        if self._stacks is None:
            self._stacks = MapperStacks(self, kinds=('mapping_',))
        return self._stacks

    def forward(self, embs):
        _require_gpu(embs, 'CleanMapper')
        names, params = [], []
        for k, p in self.named_parameters():
            names.append(k)
            params.append(p)
        return _CleanMapperFn.apply(embs, self.stacks(), names, *params)


class _MapperFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, T, names, num_words, *params):
        _require_gpu(tok, 'Mapper')
        P = dict(zip(names, [p.detach() for p in params]))
        out, saved = mapper_fwd(tok.contiguous(), T, P, num_words)
        ctx.names, ctx.P, ctx.saved, ctx.num_words = names, P, saved, num_words
        return out

    @staticmethod
    def backward(ctx, dout):
        G = mapper_bwd(dout, ctx.P, ctx.num_words, ctx.saved)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[k] for k in ctx.names)


class Mapper(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, num_words: int):
        super().__init__()
        self.num_words = num_words

        def mlp():
            return nn.Sequential(nn.Linear(input_dim, 1280), nn.LayerNorm(1280), nn.LeakyReLU(),
                                 nn.Linear(1280, 1280), nn.LayerNorm(1280), nn.LeakyReLU(),
                                 nn.Linear(1280, 1280), nn.LayerNorm(1280), nn.LeakyReLU(),
                                 nn.Linear(1280, output_dim))
        for i in range(self.num_words):
            setattr(self, f'mapping_{i}', mlp())
            setattr(self, f'mapping_patch_{i}', mlp())

    def forward(self, embs):
        """embs: list whose first entry is the image embedding -- either the reference's token-major tensor
        [B, 1+T, D] (transposed on the device here), or the `(tokens, T)` pair ClipVisionEncoder returns
        (already channel-major, no transpose)."""
        e = embs[0]
        if isinstance(e, tuple):
            tok, T = e
        else:
            _require_gpu(e, 'Mapper')
            B, T1, D = e.shape
            T = T1 - 1
            LD = K.token_ld(T)
            tok = K.transpose_pad(e.detach().contiguous(), LD).view(B, D, LD // 32, 32)
        names, params = [], []
        for k, p in self.named_parameters():
            names.append(k)
            params.append(p)
        return _MapperFn.apply(tok, T, names, self.num_words, *params)


# ---------------------------------------------------------------------------- a30 injected cross-attention
def _ld(T):
    return (T + 31) // 32 * 32


def _to_cm(x, LD):
    """token-major [B, T, D] -> channel-major [B, D, LD/32, 32] (zero-padded columns)"""
    B, T, D = x.shape
    return K.transpose_pad(x.contiguous(), LD).view(B, D, LD // 32, 32)


def _to_tm(x, T):
    """channel-major [B, D, LD/32, 32] -> token-major [B, T, D]"""
    B, D = x.shape[0], x.shape[1]
    LD = x.shape[2] * x.shape[3]
    return K.transpose_pad(x.reshape(B, D, LD), D)[:, :T]


def _lin_nobias_bwd(dout, x, w, need_dx=True):
    Cout, Cin = w.shape
    gw = K.conv_wgrad(x, dout, Cout, Cin, 1).view(Cout, Cin)
    dx = None
    if need_dx:
        wp, mp, *_ = K.pack_weights(w.view(Cout, Cin, 1, 1), PACK_DGRAD_S1)
        dx = K.conv_forward(dout, wp, mp, Cin, 1)
    return dx, gw


class CrossAttentionFn(torch.autograd.Function):
    """`inj_forward_crossattention` (main_train_i2t_mapping.py:197-233): q = to_q(hidden); k, v = to_k_global /
    to_v_global (context) when a context is given, to_k / to_v (hidden) otherwise; heads split (:85-98);
    softmax(q k^T * scale) v; to_out[0] (Linear with bias; to_out[1] is Dropout(0))."""

    @staticmethod
    def forward(ctx, hidden, context, wq, wk, wv, wo, bo, heads, scale):
        _require_gpu(hidden, 'cross_attention')
        B, Tq, _ = hidden.shape
        LDq = _ld(Tq)
        hcm = _to_cm(hidden.detach(), LDq)
        if context is None:
            ccm, Tk = hcm, Tq
        else:
            Tk = context.shape[1]
            ccm = _to_cm(context.detach(), _ld(Tk))
        q = _lin_fwd(hcm, wq.detach(), None)
        k = _lin_fwd(ccm, wk.detach(), None)
        v = _lin_fwd(ccm, wv.detach(), None)
        a, lse = K.cross_attention_fwd(q, k, v, heads, scale, Tq, Tk)
        o = _lin_fwd(a, wo.detach(), bo.detach())
        ctx.save_for_backward(hcm, ccm, q, k, v, a, lse, wq, wk, wv, wo)
        ctx.meta = (heads, scale, Tq, Tk, context is None)
        return _to_tm(o, Tq).contiguous()

    @staticmethod
    def backward(ctx, dout):
        hcm, ccm, q, k, v, a, lse, wq, wk, wv, wo = ctx.saved_tensors
        heads, scale, Tq, Tk, self_attn = ctx.meta
        do = _to_cm(dout.contiguous(), hcm.shape[2] * hcm.shape[3])
        da, gwo, gbo = _lin_bwd(do, a, wo.detach(), True)
        dq, dk, dv = K.cross_attention_bwd(q, k, v, a, da, lse, heads, scale, Tq, Tk)
        dh, gwq = _lin_nobias_bwd(dq, hcm, wq.detach())
        dc1, gwk = _lin_nobias_bwd(dk, ccm, wk.detach())
        dc2, gwv = _lin_nobias_bwd(dv, ccm, wv.detach())
        dc = K.add_(dc1, dc2)
        if self_attn:
            dh = K.add_(dh, dc)
            dctx = None
        else:
            dctx = _to_tm(dc, Tk).contiguous()
        return _to_tm(dh, Tq).contiguous(), dctx, gwq, gwk, gwv, gwo, gbo, None, None


def cross_attention(P, hidden, context, heads, scale):
    """P: dict with the attention module's parameters (to_q / to_k / to_v / to_k_global / to_v_global `.weight`,
    to_out.0.weight / .bias); hidden [B,Tq,Dq], context [B,Tk,Dc] or None -- the tensors the reference function sees."""
    kn, vn = ('to_k', 'to_v') if context is None else ('to_k_global', 'to_v_global')
    return CrossAttentionFn.apply(hidden, context, P['to_q.weight'], P[kn + '.weight'], P[vn + '.weight'],
                                  P['to_out.0.weight'], P['to_out.0.bias'], heads, scale)

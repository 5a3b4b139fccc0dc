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


LANES = int(__import__('os').environ.get('TDR_MAPPER_LANES', '4'))   # HIP streams the 2 x num_words independent MLPs are spread over


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

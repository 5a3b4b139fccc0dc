"""GPU parity of the stage-A (image-to-text mapping) pieces, SURVEY 8a rows a28-a30, through the C ABI against the
golden fixtures (tests/golden/i2t_*.npz: transformers' CLIPVisionModel; the reference's Mapper /
inj_forward_crossattention definitions) and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import i2t_oracle as IO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', params=['bx3', 'f32', 'hx2'])
def K(request):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from textualdegremoval_amd import kernels
    prev = kernels.MATH
    kernels.set_math(request.param)
    yield kernels
    kernels.set_math(prev)


def gold(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def token_major(tok, Tn):
    """[B, D, LD/32, 32] channel-major -> [B, 1+T, D] (host-side view change for comparison only)"""
    B, D = tok.shape[0], tok.shape[1]
    return tok.reshape(B, D, -1)[:, :, :Tn + 1].permute(0, 2, 1).contiguous()


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_clip_vision_encoder_vs_transformers_golden(K, tag):
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    g = gold('i2t_clip')
    hidden, inter, layers, heads, image = [int(v) for v in g[tag + '_cfg']]
    sd = IO.synth_clip_params(hidden, inter, layers, 14, image, seed=ord(tag))
    enc = ClipVisionEncoder({'vision_model.' + k: v for k, v in sd.items()}, 'cuda', heads, act=str(g[tag + '_act']))
    tok, Tn = enc.tokens(T(g[tag + '_x']).cuda())
    assert Tn == (image // 14) ** 2
    assert maxdiff(token_major(tok, Tn), T(g[tag + '_out'])) < 1e-4


@pytest.mark.parametrize('tag', ['L', 'H'])
def test_clip_full_geometry_vs_transformers_golden(K, tag):
    """BASELINE configs[3]'s encoders at full width (ViT-L/14: 1024 / 16 heads / MLP 4096 / quick_gelu; ViT-H/14: 1280 / 16
    heads of 80 / MLP 5120 / gelu), 4 layers, 224x224 -> 257 tokens, batch 2: against transformers' CLIPVisionModel as
    installed in the build container (tests/golden/make_golden_i2t.py::clip_full_geometry_cases; weights and image are
    regenerated here from the same seeds).  "Parity unpinned" against the reference's transformers 4.31.0, which is not
    vendored.  Token magnitudes reach ~300 (no post-LayerNorm, random weights), hence the relative tolerance."""
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    g = gold('i2t_clip_full')
    hidden, inter, layers, heads, image = [int(v) for v in g[tag + '_cfg']]
    sd = IO.synth_clip_params(hidden, inter, layers, 14, image, seed=ord(tag))
    enc = ClipVisionEncoder({'vision_model.' + k: v for k, v in sd.items()}, 'cuda', heads, act=str(g[tag + '_act']))
    x = torch.rand(2, 3, image, image, generator=torch.Generator().manual_seed(200 + ord(tag)))
    tok, Tn = enc.tokens(x.cuda())
    assert Tn == 256
    out = token_major(tok, Tn).cpu()
    assert out.shape == (2, 257, hidden)
    ref = T(g[tag + '_sample'])
    scale = float(g[tag + '_stats'][3])
    assert maxdiff(out[:, ::8, ::4], ref) < 2e-5 * scale
    st = np.array([out.double().mean().item(), out.double().abs().mean().item(), out.double().std().item(), out.double().abs().max().item()])
    assert np.allclose(st, g[tag + '_stats'], rtol=0, atol=2e-5 * scale)


def test_clip_encode_resizes_like_the_reference_call(K):
    """F.interpolate(image, (S, S), 'bilinear') then the encoder (main_train_i2t_mapping.py:726-730)."""
    import torch.nn.functional as F
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    sd = IO.synth_clip_params(64, 128, 1, 14, 56, seed=11)
    enc = ClipVisionEncoder(sd, 'cuda', 4)
    img = torch.rand(2, 3, 90, 90, generator=torch.Generator().manual_seed(2))
    tok, Tn = enc.encode(img.cuda(), size=56)
    ref = IO.clip_vision_tokens(sd, F.interpolate(img, (56, 56), mode='bilinear'), 4)
    assert maxdiff(token_major(tok, Tn), ref) < 1e-4


def test_glue_kernels(K):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(3, 50, 7, generator=gen)
    assert torch.equal(K.leaky_relu_fwd(x.cuda()).cpu(), F.leaky_relu(x, 0.01))
    go = torch.randn(3, 50, 7, generator=gen)
    y = F.leaky_relu(x, 0.01)
    assert torch.equal(K.leaky_relu_bwd(go.cuda(), y.cuda()).cpu(), torch.where(x > 0, go, go * 0.01))
    t = torch.randn(2, 17, 40, generator=gen)
    tp = K.transpose_pad(t.cuda(), 32).cpu()
    assert torch.equal(tp[:, :, :17], t.transpose(1, 2)) and tp[:, :, 17:].abs().max() == 0


def test_mapper_vs_reference_golden(K):
    from textualdegremoval_amd.i2t import Mapper
    g = gold('i2t_mapper')
    din, dout, words, B, Tn = [int(v) for v in g['cfg']]
    P = IO.synth_mapper_params(din, 1280, dout, words, seed=5)
    mp = Mapper(input_dim=din, output_dim=dout, num_words=words).cuda()
    assert sorted(mp.state_dict().keys()) == sorted(P.keys())
    mp.load_state_dict(P, strict=True)
    out = mp([T(g['emb']).cuda()])
    assert maxdiff(out, T(g['out'])) < 1e-4
    (out * T(g['go']).cuda()).sum().backward()
    sdp = dict(mp.named_parameters())
    for i, k in enumerate(str(n) for n in g['names']):
        gr = sdp[k].grad
        assert abs(gr.double().norm().item() - g['grad_norm'][i]) < 3e-3 * g['grad_norm'][i] + 1e-6, k
        s = gr.reshape(-1)
        smp = s[::max(1, s.numel() // 8)][:8].cpu().numpy()
        assert np.abs(smp - g['grad_sample'][i][:len(smp)]).max() < 3e-4 * max(1e-3, g['grad_norm'][i]) + 1e-6, k


def test_mapper_on_channel_major_clip_tokens_matches_oracle(K):
    """encoder -> mapper without leaving the channel-major layout (the (tokens, T) hand-over)."""
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    from textualdegremoval_amd.i2t import Mapper
    sd = IO.synth_clip_params(64, 128, 1, 14, 56, seed=4)
    enc = ClipVisionEncoder(sd, 'cuda', 4)
    x = torch.rand(3, 3, 56, 56, generator=torch.Generator().manual_seed(8))
    P = IO.synth_mapper_params(64, 1280, 24, 2, seed=6)
    mp = Mapper(64, 24, 2).cuda()
    mp.load_state_dict(P)
    out = mp([enc.tokens(x.cuda())])
    ref = IO.mapper_forward(P, IO.clip_vision_tokens(sd, x, 4), 2)
    assert maxdiff(out, ref) < 1e-4


def test_mapper_forward_backward_replayed_as_hipgraph_matches_eager(K):
    """the Mapper's forward + backward (40 small MLP chains on four stream lanes) captured once and replayed as one hipGraph, the
    way a trainer's captured step runs it: outputs and every parameter gradient equal the eager ones on fresh inputs"""
    from textualdegremoval_amd import kernels as KK
    from textualdegremoval_amd.i2t import Mapper
    P = IO.synth_mapper_params(64, 1280, 24, 6, seed=9)
    mp = Mapper(64, 24, 6).cuda()
    mp.load_state_dict(P)
    gen = torch.Generator().manual_seed(3)
    emb = torch.randn(2, 9, 64, generator=gen).cuda()
    go = torch.randn(2, 6, 24, generator=gen).cuda()

    def step():
        for p in mp.parameters():
            p.grad = None
        out = mp([emb])
        (out * go).sum().backward()
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g, refs = torch.cuda.CUDAGraph(), []
    with torch.cuda.graph(g, capture_error_mode='thread_local'), KK.workspace_capture(refs):
        out_g = step()
    grads_g = {k: p.grad for k, p in mp.named_parameters()}
    emb.copy_(torch.randn(2, 9, 64, generator=gen).cuda())      # new inputs in the captured buffers
    go.copy_(torch.randn(2, 6, 24, generator=gen).cuda())
    g.replay()
    torch.cuda.synchronize()
    got_out = out_g.clone()
    got = {k: v.clone() for k, v in grads_g.items()}
    want_out = step()
    torch.cuda.synchronize()
    assert maxdiff(got_out, want_out) == 0
    for k, p in mp.named_parameters():
        assert maxdiff(got[k], p.grad) <= 1e-6 * max(1.0, p.grad.abs().max().item()), k


@pytest.mark.parametrize('tag', ['x', 's'])
def test_injected_cross_attention_vs_reference_golden(K, tag):
    """inj_forward_crossattention forward + all gradients (hidden, context, the five weight tensors)."""
    from textualdegremoval_amd.i2t import cross_attention
    g = gold('i2t_xattn')
    dq, dc, inner, heads, B, Tq, Tk = [int(v) for v in g[tag + '_cfg']]
    P = {k[len(tag) + 3:]: T(g[k]).cuda().requires_grad_(True) for k in g.files if k.startswith(tag + '_p_')}
    hid = T(g[tag + '_hid']).cuda().requires_grad_(True)
    ctx = T(g[tag + '_ctx']).cuda().requires_grad_(True) if Tk else None
    out = cross_attention(P, hid, ctx, heads, (inner // heads) ** -0.5)
    assert maxdiff(out, T(g[tag + '_out'])) < 1e-4
    (out * T(g[tag + '_go']).cuda()).sum().backward()
    assert maxdiff(hid.grad, T(g[tag + '_ghid'])) < 1e-4
    if Tk:
        assert maxdiff(ctx.grad, T(g[tag + '_gctx'])) < 1e-4
    for k, p in P.items():
        key = f'{tag}_g_{k}'
        if key in g.files:
            assert maxdiff(p.grad, T(g[key])) < 2e-4 * max(1.0, np.abs(g[key]).max()), k


@pytest.mark.parametrize('hd,Tq,Tk', [(16, 70, 33), (64, 200, 77), (80, 130, 140)])
def test_cross_attention_kernels_vs_torch(K, hd, Tq, Tk):
    """the attention core alone (forward, lse, dq / dk / dv) at ragged lengths and every supported head dim."""
    gen = torch.Generator().manual_seed(hd + Tq)
    B, heads = 2, 2
    C = heads * hd
    LDq, LDk = (Tq + 31) // 32 * 32, (Tk + 31) // 32 * 32
    q = torch.randn(B, C, Tq, generator=gen, requires_grad=True)
    k = torch.randn(B, C, Tk, generator=gen, requires_grad=True)
    v = torch.randn(B, C, Tk, generator=gen, requires_grad=True)
    scale = hd ** -0.5
    sh = lambda z: z.reshape(B, heads, hd, -1)
    att = torch.softmax(torch.einsum('bhdq,bhdk->bhqk', sh(q), sh(k)) * scale, dim=-1)
    out = torch.einsum('bhqk,bhdk->bhdq', att, sh(v)).reshape(B, C, Tq)
    go = torch.randn(B, C, Tq, generator=gen)
    out.backward(go)

    def pad(z, LD):
        o = torch.zeros(B, C, LD)
        o[:, :, :z.shape[2]] = z.detach()
        return o.cuda().view(B, C, LD // 32, 32)
    qd, kd, vd = pad(q, LDq), pad(k, LDk), pad(v, LDk)
    o, lse = K.cross_attention_fwd(qd, kd, vd, heads, scale, Tq, Tk)
    assert maxdiff(o.reshape(B, C, LDq)[:, :, :Tq], out) < 2e-5
    assert o.reshape(B, C, LDq)[:, :, Tq:].abs().max().item() == 0
    dq, dk, dv = K.cross_attention_bwd(qd, kd, vd, o, pad(go, LDq), lse, heads, scale, Tq, Tk)
    assert maxdiff(dq.reshape(B, C, LDq)[:, :, :Tq], q.grad) < 5e-5
    assert maxdiff(dk.reshape(B, C, LDk)[:, :, :Tk], k.grad) < 5e-5
    assert maxdiff(dv.reshape(B, C, LDk)[:, :, :Tk], v.grad) < 5e-5
    assert dq.reshape(B, C, LDq)[:, :, Tq:].abs().max().item() == 0 and dk.reshape(B, C, LDk)[:, :, Tk:].abs().max().item() == 0


def test_clip_batch_flattened_layout_equals_per_image_layout(K):
    """ClipVisionEncoder.tokens(flat=True): the tokens of all images along ONE pixel axis ([1, D, B*LD/32, 32]) -- the same
    per-pixel contractions in the same order, so the result must equal the per-image layout bit for bit"""
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    sd = IO.synth_clip_params(160, 320, 2, 14, 56, seed=3)            # head dim 80 (ViT-H geometry)
    enc = ClipVisionEncoder(sd, 'cuda', 2, act='gelu')
    x = torch.rand(3, 3, 56, 56, generator=torch.Generator().manual_seed(1)).cuda()
    a, Tn = enc.tokens(x)
    f, Tf = enc.tokens(x, flat=True)
    B, D, LD = 3, 160, a.shape[2] * a.shape[3]
    assert Tn == Tf and f.shape == (1, D, B * LD // 32, 32)
    back = f.reshape(D, B, LD).permute(1, 0, 2)[:, :, :Tn + 1]
    assert torch.equal(back, a.reshape(B, D, LD)[:, :, :Tn + 1])


def test_grouped_mapper_equals_the_per_word_chains(K):
    """i2t.mapper_fwd_grouped / mapper_bwd_grouped (G-way grouped GEMMs, fused per-word LayerNorm + LeakyReLU, batch-flattened tokens)
    against the per-word launches of mapper_fwd / mapper_bwd: same arithmetic per element up to the summation order of the reductions"""
    from textualdegremoval_amd import i2t
    torch.manual_seed(4)
    words, B, Tn, din = 3, 3, 16, 64
    mp = i2t.Mapper(din, 1024, words).cuda()
    P = {k: p.data for k, p in mp.named_parameters()}
    LD = K.token_ld(Tn)
    emb = torch.randn(B, 1 + Tn, din, generator=torch.Generator().manual_seed(2)).cuda()
    tok = K.transpose_pad(emb, LD).view(B, din, LD // 32, 32)
    flat = tok.reshape(B, din, LD).permute(1, 0, 2).reshape(1, din, B * LD // 32, 32).contiguous()
    out0, sv0 = i2t.mapper_fwd(tok, Tn, P, words)
    go = torch.randn(out0.shape, generator=torch.Generator().manual_seed(3)).cuda()
    G0 = i2t.mapper_bwd(go, P, words, sv0)
    st = i2t.MapperStacks(mp)
    out1, sv1 = i2t.mapper_fwd_grouped(flat, B, Tn, st)
    G1 = i2t.mapper_bwd_grouped(go, st, sv1)
    assert maxdiff(out1, out0) < 2e-5 * max(1.0, out0.abs().max().item())
    assert sorted(G0) == sorted(G1)
    for k in G0:
        assert maxdiff(G1[k], G0[k]) <= 2e-4 * G0[k].abs().max().item() + 1e-9, k
    # the parameters now live in the stacks: an in-place update of one is visible through the other
    p = dict(mp.named_parameters())['mapping_patch_1.3.weight']
    p.data.add_(1.0)
    assert torch.equal(st.W['mapping_patch_', 3][1], p.data)


def test_clip_split_k_linears_match_the_single_launch(K):
    """the narrow-output Linears of the flat-layout encoder run split-K (K chunks as the images of one launch + tdr_splitk_finish):
    same products, partial sums added in a fixed order -- equal to the single-launch result up to fp32 summation order"""
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    sd = IO.synth_clip_params(1024, 2048, 2, 14, 56, seed=9)          # cin 1024 / 2048 -> cout 1024: the split-K rule applies
    x = torch.rand(2, 3, 56, 56, generator=torch.Generator().manual_seed(1)).cuda()
    prev = ClipVisionEncoder.SPLITK
    try:                                                                # (the channel-major engines: clip_vision.TOK16 is off by default)
        ClipVisionEncoder.SPLITK = 4
        a = ClipVisionEncoder(sd, 'cuda', 16)
        assert a.W['encoder.layers.0.out'][3] is not None and a.W['encoder.layers.0.mlp.fc2'][3] is not None and a.W['encoder.layers.0.qkv'][3] is None
        ya, T1 = a.tokens(x, flat=True)
        ClipVisionEncoder.SPLITK = 1
        b = ClipVisionEncoder(sd, 'cuda', 16)
        yb, _ = b.tokens(x, flat=True)
    finally:
        ClipVisionEncoder.SPLITK = prev
    ref = IO.clip_vision_tokens(sd, x.cpu(), 16)
    scale = ref.abs().max().item()
    assert maxdiff(ya, yb) < 1e-5 * scale
    LD = ya.shape[2] * ya.shape[3] // 2
    tm = ya.reshape(1024, 2, LD)[:, :, :T1 + 1].permute(1, 2, 0)
    assert maxdiff(tm, ref) < 1e-4 * max(1.0, scale)


def test_tok16x2_kernels_against_torch(K):
    """csrc/tdr_tok16.hip on 2-way split planes: LayerNorm -> planes, the GEMM with its three epilogues, channel-major -> planes,
    against fp64 torch on the fp32 values.  fp32-faithful: each operand carries 22 significand bits, so the bar is a few 2^-22 of the
    accumulated magnitude (sum |x||w|), not fp16 precision."""
    if K.MATH != 'hx2':
        pytest.skip('the token-major planes are the hx2 arithmetic')
    g = torch.Generator().manual_seed(8)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    P, D = 2 * 64 + 24, 1280                                             # a ragged last 64-row tile
    t, w, b = r(P, D) * 2 + 0.5, r(D), r(D)
    pl = K.tok_layernorm(t, w, b, 1e-5, planes=True)
    ref = torch.nn.functional.layer_norm(t.double(), (D,), w.double(), b.double(), 1e-5)
    assert pl.shape == (2, P, D) and maxdiff(pl[0].double() + pl[1].double(), ref) < 2e-6 * ref.abs().max().item()
    a = r(96, 72)
    ap = K.cm_to_tok16x2(a)
    assert maxdiff(ap[0].double() + ap[1].double(), a.t().double()) < 2 ** -21 * a.abs().max().item()
    for N, Kd in ((384, 1280), (1280, 160)):
        x, wt, bias = r(P, Kd), r(N, Kd) * 0.05, r(N)
        x2, w2 = K.split_planes(x), K.split_planes(wt)
        acc = x.double() @ wt.double().t() + bias.double()
        bar = 6 * 2 ** -22 * (x.abs().double() @ wt.abs().double().t()).max().item()
        assert maxdiff(K.tok16x2_gemm(x2, w2, bias, epi=3), acc.t()) < bar
        res = r(P, N)
        assert maxdiff(K.tok16x2_gemm(x2, w2, bias, epi=2, out32=res.clone()), res.double() + acc) < bar + 1e-6
        for act, f in ((0, lambda v: v), (2, torch.nn.functional.gelu), (3, lambda v: v * torch.sigmoid(1.702 * v))):
            y = K.tok16x2_gemm(x2, w2, bias, epi=4, act=act)
            assert maxdiff(y[0].double() + y[1].double(), f(acc)) < 2 * bar + 2 ** -21 * acc.abs().max().item(), act
        y = K.tok16x2_gemm(x2, w2, None, epi=3)
        assert maxdiff(y, (acc - bias.double()).t()) < bar


@pytest.mark.parametrize('hidden,inter,heads,act', [(1024, 2048, 16, 'quick_gelu'), (1280, 2560, 16, 'gelu')])
def test_clip_token_major_planes_match_the_channel_major_engines(K, hidden, inter, heads, act):
    """ClipVisionEncoder.tokens(flat=True) under hx2 at ViT-L / ViT-H widths: blocks on tdr_tok16x2_gemm (operands pre-split into
    hi | lo fp16 planes, token-major) against the oracle and against the channel-major engines -- one arithmetic, two layouts"""
    if K.MATH != 'hx2':
        pytest.skip('the token-major planes are the hx2 arithmetic')
    from textualdegremoval_amd.clip_vision import ClipVisionEncoder
    sd = IO.synth_clip_params(hidden, inter, 2, 14, 56, seed=hidden)
    x = torch.rand(3, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    from textualdegremoval_amd import clip_vision as _cv
    _cv.TOK16 = True                                                    # opt-in path (measured neutral on the stage-A step)
    try:
        enc = ClipVisionEncoder(sd, 'cuda', heads, act=act)
    finally:
        _cv.TOK16 = False
    assert enc.tok16
    f, Tn = enc.tokens(x.cuda(), flat=True)
    c, _ = enc.tokens(x.cuda())                                          # per-image layout: the channel-major engines
    ref = IO.clip_vision_tokens(sd, x, heads, act=act)
    scale = ref.abs().max().item()
    LD = c.shape[2] * c.shape[3]
    back = f.reshape(hidden, 3, LD).permute(1, 0, 2)[:, :, :Tn + 1]
    assert maxdiff(back, c.reshape(3, hidden, LD)[:, :, :Tn + 1]) < 2e-5 * scale
    assert maxdiff(back.permute(0, 2, 1), ref) < 2e-5 * scale

"""Generate golden vectors for the stage-A (image-to-text mapping) pieces, SURVEY.md 8a rows a28-a30.

Run in the build container only:
    python tests/golden/make_golden_i2t.py
Writes tests/golden/i2t_*.npz (data only; weights are regenerated from seeds by oracle.i2t_oracle.synth_*).

  a28  CLIP ViT image encoder: THIRD-PARTY (`transformers`, pinned 4.31.0 by the reference, not vendored).  The vectors
       come from `transformers.CLIPVisionModel` as installed here (version recorded in the file) with random weights,
       called the way the reference calls it (scripts/train/main_train_i2t_mapping.py:726-731).
  a29/a30  `Mapper`, `inj_forward_crossattention` and the head reshapes: the reference defines them inside
       scripts/train/main_train_i2t_mapping.py, whose module-level imports need `diffusers` (absent).  Their
       definitions are therefore located with `ast` and executed from the reference file in a namespace holding
       torch only -- the reference's code runs, nothing of it is copied; only numeric outputs are saved.
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SCRIPT = '/root/reference/scripts/train/main_train_i2t_mapping.py'

from oracle import i2t_oracle as IO  # noqa: E402


def sample(t, n=16):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def load_reference_defs(names, script=None):
    src = open(script or REF_SCRIPT).read()
    tree = ast.parse(src)
    import types
    from typing import Optional, Tuple, Union
    ns = {'torch': torch, 'nn': nn, 'F': F, 'Optional': Optional, 'Tuple': Tuple, 'Union': Union,
          'BaseModelOutputWithPooling': lambda **kw: types.SimpleNamespace(**kw)}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            node.decorator_list = []
            exec(compile(ast.Module([node], []), REF_SCRIPT, 'exec'), ns)
    return ns


def clip_cases():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    d = {'transformers_version': np.array(transformers.__version__)}
    for tag, (hidden, inter, layers, heads, image, act) in {
            'a': (64, 128, 2, 4, 56, 'quick_gelu'),          # head dim 16, 16 patch tokens
            'b': (160, 320, 1, 2, 42, 'gelu')}.items():      # head dim 80 (ViT-H geometry), 9 patch tokens
        cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                               num_attention_heads=heads, image_size=image, patch_size=14, hidden_act=act)
        m = CLIPVisionModel(cfg).eval()
        sd = IO.synth_clip_params(hidden, inter, layers, 14, image, seed=ord(tag))
        keys = list(m.state_dict().keys())
        pref = 'vision_model.' if keys[0].startswith('vision_model.') else ''
        missing = m.load_state_dict({pref + k: v for k, v in sd.items()}, strict=False)
        assert not [k for k in missing.missing_keys if 'position_ids' not in k], missing
        g = torch.Generator().manual_seed(100 + ord(tag))
        big = torch.rand(2, 3, 70, 70, generator=g)
        x = F.interpolate(big, (image, image), mode='bilinear')          # :726
        with torch.no_grad():
            feats = m(x, output_hidden_states=True)
        d[f'{tag}_x'] = x.numpy()
        d[f'{tag}_out'] = feats[0].numpy()                               # image_features[0] (:730)
        d[f'{tag}_cfg'] = np.array([hidden, inter, layers, heads, image])
        d[f'{tag}_act'] = np.array(act)
        print('clip', tag, feats[0].shape, float(feats[0].abs().mean()))
    np.savez_compressed(os.path.join(HERE, 'i2t_clip.npz'), **d)


def clip_full_geometry_cases():
    """BASELINE configs[3]'s encoders at their FULL width, 4 of their layers: CLIP ViT-L/14 (hidden 1024, 16 heads x 64,
    MLP 4096, quick_gelu) and ViT-H/14 (hidden 1280, 16 heads x 80, MLP 5120, gelu) on a 224x224 image (257 tokens).
    Weights (IO.synth_clip_params) and the image are regenerated from seeds by the test, so the fixture holds only a
    strided sample of the output tokens and their statistics.  Pinned to the transformers version installed here
    (the reference pins 4.31.0, requirements.txt:2 -- not vendored: parity unpinned against that exact version)."""
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    d = {'transformers_version': np.array(transformers.__version__)}
    for tag, (hidden, inter, layers, heads, image, act) in {'L': (1024, 4096, 4, 16, 224, 'quick_gelu'),
                                                            'H': (1280, 5120, 4, 16, 224, 'gelu')}.items():
        cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                               num_attention_heads=heads, image_size=image, patch_size=14, hidden_act=act)
        m = CLIPVisionModel(cfg).eval()
        sd = IO.synth_clip_params(hidden, inter, layers, 14, image, seed=ord(tag))
        keys = list(m.state_dict().keys())
        pref = 'vision_model.' if keys[0].startswith('vision_model.') else ''
        missing = m.load_state_dict({pref + k: v for k, v in sd.items()}, strict=False)
        assert not [k for k in missing.missing_keys if 'position_ids' not in k], missing
        x = torch.rand(2, 3, image, image, generator=torch.Generator().manual_seed(200 + ord(tag)))
        with torch.no_grad():
            out = m(x, output_hidden_states=True)[0]
        d[f'{tag}_cfg'] = np.array([hidden, inter, layers, heads, image])
        d[f'{tag}_act'] = np.array(act)
        d[f'{tag}_sample'] = out[:, ::8, ::4].numpy()
        d[f'{tag}_stats'] = np.array([out.double().mean().item(), out.double().abs().mean().item(), out.double().std().item(),
                                      out.double().abs().max().item()])
        print('clip full geometry', tag, tuple(out.shape), d[f'{tag}_stats'])
    np.savez_compressed(os.path.join(HERE, 'i2t_clip_full.npz'), **d)


def mapper_case():
    ns = load_reference_defs({'Mapper'})
    din, dout, words, B, T = 48, 40, 2, 2, 16
    mp = ns['Mapper'](input_dim=din, output_dim=dout, num_words=words)
    P = IO.synth_mapper_params(din, 1280, dout, words, seed=5)
    assert sorted(P.keys()) == sorted(mp.state_dict().keys())
    mp.load_state_dict(P)
    g = torch.Generator().manual_seed(9)
    emb = torch.randn(B, 1 + T, din, generator=g)
    out = mp([emb])
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    d = dict(emb=emb.numpy(), out=out.detach().numpy(), go=go.numpy(), cfg=np.array([din, dout, words, B, T]))
    names = sorted(P.keys())
    d['names'] = np.array(names)
    sdp = dict(mp.named_parameters())
    d['grad_norm'] = np.array([sdp[k].grad.double().norm().item() for k in names])
    d['grad_sample'] = np.stack([np.pad(sample(sdp[k].grad, 8), (0, 8 - min(8, sdp[k].grad.numel()))) for k in names])
    np.savez_compressed(os.path.join(HERE, 'i2t_mapper.npz'), **d)
    print('mapper', out.shape, float(out.abs().mean()))


def clean_mapper_case():
    """`CleanMapper` (scripts/train/main_train_tr_mapping.py:84-120), executed from the reference file: forward on the words of a
    Mapper-shaped input, gradient w.r.t. every parameter and w.r.t. the input words."""
    ns = load_reference_defs({'CleanMapper'}, script='/root/reference/scripts/train/main_train_tr_mapping.py')
    din, dout, words, B = 40, 40, 3, 2
    cm = ns['CleanMapper'](input_dim=din, output_dim=dout, num_words=words)
    P = IO.synth_clean_mapper_params(din, 1280, dout, words, seed=6)
    assert sorted(P.keys()) == sorted(cm.state_dict().keys())
    cm.load_state_dict(P)
    g = torch.Generator().manual_seed(10)
    inj = torch.randn(B, words, din, generator=g, requires_grad=True)
    out = cm(inj)
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    names = list(cm.state_dict().keys())                                      # registration order of the reference class
    sdp = dict(cm.named_parameters())
    d = dict(inj=inj.detach().numpy(), out=out.detach().numpy(), go=go.numpy(), dinj=inj.grad.numpy(), cfg=np.array([din, dout, words, B]),
             names=np.array(names), grad_norm=np.array([sdp[k].grad.double().norm().item() for k in names]),
             grad_sample=np.stack([np.pad(sample(sdp[k].grad, 8), (0, 8 - min(8, sdp[k].grad.numel()))) for k in names]))
    np.savez_compressed(os.path.join(HERE, 'i2t_clean_mapper.npz'), **d)
    print('clean mapper', out.shape, float(out.abs().mean()))


def cross_attention_case():
    ns = load_reference_defs({'inj_forward_crossattention', 'reshape_heads_to_batch_dim', 'reshape_batch_dim_to_heads'})

    class Attn(nn.Module):          # the attributes diffusers' CrossAttention carries (:197-233 reads exactly these)
        def __init__(self, dq, dc, inner, heads):
            super().__init__()
            self.heads, self.scale = heads, (inner // heads) ** -0.5
            self.to_q = nn.Linear(dq, inner, bias=False)
            self.to_k = nn.Linear(dq, inner, bias=False)
            self.to_v = nn.Linear(dq, inner, bias=False)
            self.to_k_global = nn.Linear(dc, inner, bias=False)
            self.to_v_global = nn.Linear(dc, inner, bias=False)
            self.to_out = nn.ModuleList([nn.Linear(inner, dq), nn.Dropout(0.0)])
    Attn.reshape_heads_to_batch_dim = ns['reshape_heads_to_batch_dim']
    Attn.reshape_batch_dim_to_heads = ns['reshape_batch_dim_to_heads']
    Attn.forward = ns['inj_forward_crossattention']
    d = {}
    for tag, (dq, dc, inner, heads, B, Tq, Tk) in {'x': (48, 40, 64, 2, 2, 96, 77), 's': (64, 40, 128, 2, 1, 80, 0)}.items():
        torch.manual_seed(3 + ord(tag))
        at = Attn(dq, dc, inner, heads)
        g = torch.Generator().manual_seed(ord(tag))
        hid = torch.randn(B, Tq, dq, generator=g, requires_grad=True)
        ctx = torch.randn(B, Tk, dc, generator=g, requires_grad=True) if Tk else None
        out = at(hid, {'CONTEXT_TENSOR': ctx} if Tk else None)
        go = torch.randn(out.shape, generator=g)
        (out * go).sum().backward()
        d.update({f'{tag}_hid': hid.detach().numpy(), f'{tag}_out': out.detach().numpy(), f'{tag}_go': go.numpy(),
                  f'{tag}_ghid': hid.grad.numpy(), f'{tag}_cfg': np.array([dq, dc, inner, heads, B, Tq, Tk])})
        if Tk:
            d[f'{tag}_ctx'] = ctx.detach().numpy(); d[f'{tag}_gctx'] = ctx.grad.numpy()
        for k, p in at.named_parameters():
            d[f'{tag}_p_{k}'] = p.detach().numpy()
            if p.grad is not None:
                d[f'{tag}_g_{k}'] = p.grad.numpy()
        print('xattn', tag, out.shape)
    np.savez_compressed(os.path.join(HERE, 'i2t_xattn.npz'), **d)


def text_injection_case():
    """`inj_forward_text` (:108-194), the patched CLIPTextTransformer.__call__: the reference's own function is executed on a
    stand-in `self` that carries what it reads -- embeddings.token_embedding, embeddings(...) (adds the position embedding to
    inputs_embeds), encoder (ONE Linear here, the stage-A stand-in for the 23 third-party transformer layers), final_layer_norm
    and config.  The placeholder-token injection (:139-151) and the final LayerNorm (:176) are the reference's lines."""
    import types
    from typing import Optional, Tuple, Union
    ns = load_reference_defs({'inj_forward_text', '_build_causal_attention_mask'})
    fn = ns['inj_forward_text']
    V, D, S, L, B = 50, 24, 77, 5, 3
    g = torch.Generator().manual_seed(21)

    class Emb(nn.Module):
        def __init__(self):
            super().__init__()
            self.token_embedding = nn.Embedding(V, D)
            self.position_embedding = nn.Embedding(S, D)

        def forward(self, input_ids=None, position_ids=None, inputs_embeds=None):
            return inputs_embeds + self.position_embedding.weight[None, :inputs_embeds.shape[1]]

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(D, D)

        def forward(self, inputs_embeds=None, **kw):
            return (self.proj(inputs_embeds),)
    me = types.SimpleNamespace(embeddings=Emb(), encoder=Enc(), final_layer_norm=nn.LayerNorm(D),
                               config=types.SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=False))
    with torch.no_grad():
        for p in list(me.embeddings.parameters()) + list(me.encoder.parameters()) + list(me.final_layer_norm.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.2) + (1.0 if p.dim() == 1 and p is me.final_layer_norm.weight else 0.0))
    ids = torch.randint(0, V, (B, S), generator=g)
    inj = torch.randn(B, L, D, generator=g, requires_grad=True)
    idx = torch.tensor([1, 40, S - L])                          # first slot after BOS, middle, last position that still fits
    out = fn(me, {'input_ids': ids, 'inj_embedding': inj, 'inj_index': idx})[0]
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    d = dict(ids=ids.numpy(), inj=inj.detach().numpy(), idx=idx.numpy(), out=out.detach().numpy(), go=go.numpy(), ginj=inj.grad.numpy(),
             tok=me.embeddings.token_embedding.weight.detach().numpy(), pos=me.embeddings.position_embedding.weight.detach().numpy(),
             proj_w=me.encoder.proj.weight.detach().numpy(), proj_b=me.encoder.proj.bias.detach().numpy(),
             ln_w=me.final_layer_norm.weight.detach().numpy(), ln_b=me.final_layer_norm.bias.detach().numpy())
    np.savez_compressed(os.path.join(HERE, 'i2t_text_inject.npz'), **d)
    print('text injection', out.shape, float(out.abs().mean()))


if __name__ == '__main__':
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == 'text_inject':
        text_injection_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'clean_mapper':
        clean_mapper_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'clip_full':
        clip_full_geometry_cases()
        sys.exit(0)
    clip_cases()
    clip_full_geometry_cases()
    mapper_case()
    clean_mapper_case()
    cross_attention_case()
    text_injection_case()

"""Golden vectors for the DINOv2 window matcher, produced by running the REFERENCE classes on CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_dino.py
Writes tests/golden/dino_*.npz: the expected outputs only; parameters and inputs are regenerated from seeds by
oracle.dino_oracle.synth_vit_params / torch generators.  Nothing of the reference is copied: it is imported,
executed and its numeric outputs are saved.
"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'
from oracle import dino_oracle as D  # noqa: E402


def ref_vit(embed, depth, heads):
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    vt = importlib.import_module('models.dino.vision_transformers')
    net = vt.DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=embed, depth=depth, num_heads=heads, mlp_ratio=4,
                                   init_values=1.0, ffn_layer='mlp', block_chunks=0)
    return net.eval()


def images(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(H // 16, 2), max(W // 16, 2), generator=g)
    return F.interpolate(low, size=(H, W), mode='bicubic').clamp(0, 1)


def main():
    torch.manual_seed(0)
    embed, depth, heads = 32, 2, 2
    sd = D.synth_vit_params(embed, depth, heads, seed=11)
    net = ref_vit(embed, depth, heads)
    missing = net.load_state_dict(sd, strict=True)
    out = {}
    # 1) patch tokens: square without pos-embed interpolation is 518 only; small inputs always interpolate
    for tag, (B, H, W) in {'sq56': (2, 56, 56), 'rect70x42': (1, 70, 42), 'sq140': (1, 140, 140)}.items():
        x = images(B, H, W, seed=100 + H)
        with torch.no_grad():
            y = net(x)                                           # forward(): x_norm_patchtokens through Identity head
        out[f'tokens_{tag}'] = y.numpy()
    # 2) window matching exactly as optimize_parameters does it (image_restoration_ref_model.py:215-247)
    B, h = 2, 48
    clean = images(B, 96, 96, seed=7)
    ref = clean
    g = torch.Generator().manual_seed(8)
    lq = clean[:, :, 24:72, 12:60] + torch.randn(B, 3, h, h, generator=g) * (15 / 255)
    stride = int(h // 4)
    with torch.no_grad():
        un = F.unfold(ref.clone(), kernel_size=(h, h), stride=(stride, stride))
        _, L, N = un.shape
        un = un.transpose(-1, -2).contiguous().view(B * N, 3, h, h)
        size = (int(math.ceil(h / 14) * 14), int(math.ceil(h / 14) * 14))
        fl = net(F.interpolate(lq.clone(), size=size, mode='bilinear')).view(B, 1, -1)
        fr = net(F.interpolate(un.clone(), size=size, mode='bilinear')).view(B, N, -1)
        corr = torch.matmul(F.normalize(fl, dim=-1), F.normalize(fr, dim=-1).transpose(-1, -2))
        _, idx = torch.topk(corr, k=1, dim=-1)
        idx = idx[:, :, 0]
        ref_in = torch.gather(un.view(B, N, 3, h, h), 1, idx[:, :, None, None, None].expand(-1, -1, 3, h, h)).squeeze(1)
    out['match_corr'] = corr[:, 0].numpy()
    out['match_index'] = idx[:, 0].numpy()
    out['match_ref_in_sum'] = np.array([ref_in.double().sum().item(), ref_in.double().abs().sum().item()])
    srt = np.sort(corr[:, 0].numpy(), axis=1)
    out['match_gap'] = srt[:, -1] - srt[:, -2]
    np.savez_compressed(os.path.join(HERE, 'dino_vit_e32_d2.npz'), **out)
    print('index', out['match_index'], 'gap', out['match_gap'])


if __name__ == '__main__':
    main()

"""CPU: the stage-A step oracle (oracle/i2t_oracle.py stage_a_loss / OracleStageATrainer) against what the reference pins:
the placeholder-token injection + final LayerNorm of `inj_forward_text` (tests/golden/i2t_text_inject.npz, produced by executing
the reference's function, make_golden_i2t.py::text_injection_case), and internal consistency of the composed step."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import i2t_oracle as IO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'i2t_text_inject.npz'))
T = lambda k: torch.from_numpy(G[k])


def test_injection_and_final_layernorm_match_inj_forward_text():
    inj = T('inj').clone().requires_grad_(True)
    x = IO.inject_embeddings(T('tok')[T('ids')], inj, T('idx')) + T('pos')
    out = F.layer_norm(F.linear(x, T('proj_w'), T('proj_b')), (x.shape[-1],), T('ln_w'), T('ln_b'), 1e-5)
    assert (out - T('out')).abs().max().item() < 2e-6
    (out * T('go')).sum().backward()
    assert (inj.grad - T('ginj')).abs().max().item() < 2e-6


def test_oracle_step_trains_every_tensor_and_clips():
    from textualdegremoval_amd import stage_a as SA          # stub / batch builders are data generators (no GPU needed)
    levels = (('lvA', 1, 64, 1), ('lvB', 4, 128, 2))
    S = SA.stage_a_stub(seed=3, vocab=50, levels=levels)
    batch = SA.synthetic_batch(2, size=128, vocab=50, num_words=3, seed=1)
    P = IO.synth_mapper_params(32, 1280, 1024, 3, seed=5)
    for name, _, _, _ in levels:
        P[name + '_to_k.weight'], P[name + '_to_v.weight'] = S[name + '.to_k.weight'].clone(), S[name + '.to_v.weight'].clone()
    tr = IO.OracleStageATrainer(P, S, None, None, None, levels, 3)
    emb = torch.randn(2, 17, 32, generator=torch.Generator().manual_seed(0))
    losses = [tr.step(batch, emb) for _ in range(3)]
    assert all(np.isfinite(losses)) and tr.last_norm > 1.0       # the clip (max-norm 1) is active
    assert all(g.abs().max() > 0 for g in tr.last_grads.values())   # every trained tensor receives a gradient

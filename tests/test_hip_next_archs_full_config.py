"""The two "next" architectures at the block counts of the reference's own YAMLs (dim 48, num_blocks [4,6,6,8], fusion [2,2,2,2],
MASA encoder [4,4,4,4]) -- 001_promptir_all_in_one_restoration.yml (decoder=True, see R4) and 008_drsformer_*.yml -- one 128x128
pair straight against the CPU oracle: every MASA match decision outside exact ties, the output (1e-4) and every parameter gradient
(5e-3 of the tensor maximum).  The small-configuration goldens pin the oracle to the reference; this pins the HIP path to the
oracle where the depth (26 / 24 transformer blocks in series) could let an error grow."""
import numpy as np
import pytest
import torch

from oracle import drsformer_ref_oracle as DO
from oracle import nafnet_ref_oracle as NO
from oracle import promptir_ref_oracle as PO

pytestmark = pytest.mark.gpu


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def _compare(engine, fwd, params, cfg, monkeypatch, unused=(), out_tol=1e-4, grad_tol=5e-3):
    """HIP first; the oracle then follows the HIP match decisions where its own scores are near-ties (as tests/test_hip_full_size.py
    does: a flipped tie moves a patch of warped features, which is a different -- equally valid -- function of the inputs)."""
    torch.set_num_threads(16)
    lq, gt, ref = NO.synth_pair(1, 128, 128, seed=2024)
    names = [k for k in params if not (unused and k.startswith(unused))]
    Pc = {k: params[k].cuda().contiguous() for k in names}
    from textualdegremoval_amd import kernels as K
    import os
    prev = K.MATH
    K.set_math(prev)
    try:
        out, saved = engine.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
        # a FIXED cotangent instead of the L1 gradient: sign(out - gt) flips wherever the residual is within rounding of zero, and
        # PromptIR's prompt components are spatial maps -- one flipped pixel is a local 2 % gradient difference there (measured)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(99)) / out.numel()
        G = engine.net_bwd(cot.cuda(), Pc, cfg, saved)
    finally:
        K.set_math(prev)
    sv_masa = saved[6]
    hip_index, hip_index_all = sv_masa[4].cpu().long(), sv_masa[7].cpu().long()
    seen = {}
    orig_cs, orig_fs = NO.coarse_search, NO.fine_search

    def cs(lrb, r4, dil):
        total, index = orig_cs(lrb, r4, dil)
        hi = hip_index.view_as(index)
        gap = (total.gather(2, index.unsqueeze(2)) - total.gather(2, hi.unsqueeze(2))).squeeze(2)
        seen['coarse'] = ((index != hi).sum().item(), index.numel(), gap.abs().max().item())
        return total, hi

    def fs(lrb_flat, refb):
        val, idx, corr = orig_fs(lrb_flat, refb)
        B = corr.shape[0]
        hi = hip_index_all.view(B, -1)
        v2 = corr.gather(2, hi.unsqueeze(2)).squeeze(2)
        gap = val.reshape(B, -1) - v2
        seen['fine'] = ((idx.reshape(B, -1) != hi).sum().item(), hi.numel(), gap.abs().max().item())
        return v2.view_as(val), hi.view_as(idx), corr

    monkeypatch.setattr(NO, 'coarse_search', cs)
    monkeypatch.setattr(NO, 'fine_search', fs)
    Pr = {k: v.clone().requires_grad_(k in names) for k, v in params.items()}
    oo = fwd(Pr, cfg, lq, ref)
    oo.backward(cot)
    print('match decisions (mismatches, total, largest oracle score gap at a mismatch):', seen)
    assert seen['coarse'][2] < 1e-5 and seen['fine'][2] < 1e-5 and seen['fine'][0] <= 0.02 * seen['fine'][1], seen
    print('max |out - oracle|', maxdiff(out, oo))
    assert maxdiff(out, oo) < out_tol
    worst, bad = 0.0, []
    for k in names:
        g = Pr[k].grad
        e = maxdiff(G[k].view_as(g), g) / max(g.abs().max().item(), 1e-12)
        worst = max(worst, e)
        if e >= grad_tol:
            bad.append((k, e))
    print('max relative gradient error over tensors', worst)
    assert not bad, bad


def test_promptir_yaml_network_vs_oracle(monkeypatch):
    from textualdegremoval_amd import promptir_engine as PE
    cfg = PO.default_cfg(num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2])
    _compare(PE, PO.promptir_ref_forward, PO.synth_params(cfg, seed=11), cfg, monkeypatch, PO.UNUSED)


def test_drsformer_yaml_network_vs_oracle():
    """DRSformer-ref's top-k attention is a DISCONTINUOUS function of its logits: with 48-96 channels per head and 54 attention
    blocks in series some row always has two logits within a few 1e-6 of each other at one of the four k boundaries (measured on
    this input: smallest gap 2.7e-6 at latent.2, 7.8e-6 at encoder_level3.3), and whichever side of it fp32 rounding falls on, the
    kept set -- and from there the rest of the network -- changes by ~1e-3.  A whole-network comparison at depth is therefore
    done block by block on the ORACLE's activations: every block of the oracle's forward pass is re-run on the HIP engine from
    the oracle's own input; blocks whose top-k boundaries are separated by more than 2e-5 must agree to 2e-5, the few fragile
    ones and the free-running network to 5e-3."""
    import torch.nn.functional as F
    from textualdegremoval_amd import drsformer_engine as DE, engine as E, kernels as K
    torch.set_num_threads(16)
    cfg = dict(DO.default_cfg(dim=48, nf=48, num_blocks=[4, 6, 6, 8], ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2]), mefc=True)
    P = DO.full_synth_params(cfg, seed=12)
    lq, gt, ref = NO.synth_pair(1, 128, 128, seed=2024)
    rec = {}
    orig_tb, orig_ms, orig_tk = DO.transformer_block, DO.mefc_subnet, DO.tksa

    def tk(x, Pd, pre, heads):
        b, c, h, w = x.shape
        t = F.conv2d(F.conv2d(x, Pd[pre + 'qkv.weight'], Pd.get(pre + 'qkv.bias')), Pd[pre + 'qkv_dwconv.weight'], Pd.get(pre + 'qkv_dwconv.bias'),
                     padding=1, groups=3 * c)
        q, k, _ = t.chunk(3, dim=1)
        ch = c // heads
        a = (F.normalize(q.reshape(b, heads, ch, h * w), dim=-1) @ F.normalize(k.reshape(b, heads, ch, h * w), dim=-1).transpose(-2, -1)) * Pd[pre + 'temperature']
        srt = a.sort(dim=-1, descending=True).values
        rec[pre[:-5]]['gap'] = min((srt[..., kk - 1] - srt[..., kk]).min().item() for kk in (int(ch / 2), int(ch * 2 / 3), int(ch * 3 / 4), int(ch * 4 / 5)))
        return orig_tk(x, Pd, pre, heads)

    def tb(x, Pd, pre, heads, ln):
        rec[pre] = {'x': x.detach().clone(), 'heads': heads}
        y = orig_tb(x, Pd, pre, heads, ln)
        rec[pre]['y'] = y.detach()
        return y

    def ms(x, Pd, pre):
        y = orig_ms(x, Pd, pre)
        rec[pre] = {'x': x.detach().clone(), 'y': y.detach(), 'mefc': True}
        return y
    DO.tksa, DO.transformer_block, DO.mefc_subnet = tk, tb, ms
    try:
        with torch.no_grad():
            oo = DO.drsformer_full_forward(P, cfg, lq, ref)
    finally:
        DO.tksa, DO.transformer_block, DO.mefc_subnet = orig_tk, orig_tb, orig_ms
    Pc = {k: v.cuda().contiguous() for k, v in P.items()}
    n_solid = n_fragile = 0
    for pre, r in rec.items():
        if r.get('mefc'):
            y, _ = DE.mefc_fwd(r['x'].cuda(), Pc, pre)
            assert maxdiff(y, r['y']) < 2e-5, pre
            continue
        y, _ = DE.tblock_fwd(r['x'].cuda().contiguous(), E._sub(Pc, pre), r['heads'], cfg['LayerNorm_type'])
        e = maxdiff(y, r['y'])
        if r['gap'] > 2e-5:
            n_solid += 1
            assert e < 2e-5, (pre, e, r['gap'])
        else:
            n_fragile += 1
            assert e < 5e-3, (pre, e, r['gap'])
    print('blocks compared on the oracle\'s activations:', n_solid, 'with separated top-k boundaries,', n_fragile, 'fragile')
    assert n_solid >= 24 and n_solid + n_fragile == 48
    out, _ = DE.net_fwd(Pc, cfg, lq.cuda(), ref.cuda())
    print('free-running network: max |out - oracle|', maxdiff(out, oo))
    assert maxdiff(out, oo) < 5e-3

"""Pins the CPU oracle (oracle/nafnet_ref_oracle.py) against golden vectors that
were produced by running the reference itself (tests/golden/make_golden.py).
Tolerances: 1e-5 max-abs on O(1) fp32 activations (reference target is 1e-4),
exact equality for integer indices."""
import os

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as O



def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_layernorm2d_fwd_bwd(golden_dir):
    g = load(golden_dir, 'per_op')
    x = T(g['ln_x']).requires_grad_(True); w = T(g['ln_w']).requires_grad_(True); b = T(g['ln_b']).requires_grad_(True)
    y = O.layernorm2d(x, w, b, 1e-6)
    y.backward(T(g['ln_go']))
    assert np.abs(y.detach().numpy() - g['ln_y']).max() < 1e-5
    assert np.abs(x.grad.numpy() - g['ln_gx']).max() < 1e-5
    assert np.abs(w.grad.numpy() - g['ln_gw']).max() < 1e-4
    assert np.abs(b.grad.numpy() - g['ln_gb']).max() < 1e-4


def test_nafblock_fwd_bwd(golden_dir):
    g = load(golden_dir, 'per_op')
    P = {str(k): T(g['naf_p_' + str(k)]).requires_grad_(True) for k in g['naf_names']}
    x = T(g['naf_x']).requires_grad_(True)
    y = O.naf_block(x, P, '')
    y.backward(T(g['naf_go']))
    assert np.abs(y.detach().numpy() - g['naf_y']).max() < 2e-5
    assert np.abs(x.grad.numpy() - g['naf_gx']).max() < 2e-5
    for k, p in P.items():
        ref = g['naf_g_' + k]
        assert np.abs(p.grad.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k


def test_masa_encoder_fwd_bwd(golden_dir):
    g = load(golden_dir, 'per_op')
    P = {'e.' + str(k): T(g['enc_p_' + str(k)]).requires_grad_(True) for k in g['enc_names']}
    x = T(g['enc_x']).requires_grad_(True)
    feats = O.masa_encoder(x, P, 'e.', [1, 1, 1, 1])
    for i, f in enumerate(feats):
        assert np.abs(f.detach().numpy() - g[f'enc_f{i}']).max() < 1e-5
    sum((f * f).mean() * (i + 1) for i, f in enumerate(feats)).backward()
    assert np.abs(x.grad.numpy() - g['enc_gx']).max() < 1e-5
    for k, p in P.items():
        assert np.abs(p.grad.numpy() - g['enc_g_' + k[2:]]).max() < 1e-5, k


def test_fine_search_value_index_and_grads(golden_dir):
    g = load(golden_dir, 'masa_ops')
    lr = T(g['so_lr']).requires_grad_(True); rf = T(g['so_ref']).requires_grad_(True)
    val, idx, _ = O.fine_search(lr, rf)
    assert np.array_equal(idx.numpy(), g['so_idx'])
    assert np.abs(val.detach().numpy()[:, 0] - g['so_val']).max() < 1e-6
    (val[:, 0] * T(g['so_go'])).sum().backward()
    assert np.abs(lr.grad.numpy() - g['so_glr']).max() < 1e-5
    assert np.abs(rf.grad.numpy() - g['so_gref']).max() < 1e-5


@pytest.mark.parametrize('s', [1, 2, 4])
def test_transfer_fused_form(golden_dir, s):
    g = load(golden_dir, 'masa_ops')
    fea = T(g[f'tr{s}_fea']).requires_grad_(True); att = T(g[f'tr{s}_att']).requires_grad_(True)
    out = O.transfer(fea, T(g[f'tr{s}_idx']), att, s, 13)
    assert np.abs(out.detach().numpy() - g[f'tr{s}_out']).max() < 1e-5
    (out * T(g[f'tr{s}_go'])).sum().backward()
    assert np.abs(fea.grad.numpy() - g[f'tr{s}_gfea']).max() < 1e-5
    assert np.abs(att.grad.numpy() - g[f'tr{s}_gatt']).max() < 2e-5


def test_coarse_search_centre_tap_form(golden_dir):
    g = load(golden_dir, 'masa_ops')
    corr, idx = O.coarse_search(T(g['cs_lr']), T(g['cs_ref']), [1, 2, 3])
    assert np.array_equal(idx.numpy(), g['cs_idx'])
    assert np.abs(corr.max(dim=2).values.numpy() - g['cs_val']).max() < 1e-5


CASES = [('net_w8_128_wrap', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_256_b2', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_120x100_pad', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_cfg1_w16_128', dict(width=16, nf=16, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])),
         # the shipped NAFNet YAML's widths (width = nf = 64: up to 2048 channels in the middle fusion block), make_golden.py yaml64
         ('net_yaml_w64_128', dict(width=64, nf=64, enc_blk_nums=[1, 1, 1, 3], dec_blk_nums=[1, 1, 1, 1], middle_blk_num=1,
                                   ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 1])),
         # ref of another size than lq (validation / inference, image_restoration_ref_model.py:286-330)
         ('net_w8_256_ref384', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_128_ref256_wrap', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])),
         ('net_w8_200x136_ref300', dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]))]
REF_HW = {'net_w8_256_ref384': (384, 384), 'net_w8_128_ref256_wrap': (256, 256), 'net_w8_200x136_ref300': (300, 300)}


@pytest.mark.parametrize('name,kw', CASES)
def test_whole_net_forward_backward(golden_dir, name, kw):
    g = load(golden_dir, name)
    cfg = O.default_cfg(**kw)
    seed = int(g['seed'])
    P = {k: v.requires_grad_(True) for k, v in O.synth_params(cfg, seed=seed).items()}
    lq, gt, ref = O.synth_pair(int(g['cfg_B']), int(g['cfg_H']), int(g['cfg_W']), seed=1234 + seed, ref_hw=REF_HW.get(name))
    out, aux = O.nafnet_ref_forward(P, cfg, lq, ref, return_aux=True)
    assert np.array_equal(aux['index'].numpy(), g['index'][..., 0] if g['index'].ndim == 3 else g['index'])
    assert np.array_equal(aux['index_all'].numpy(), g['index_all'])
    assert np.abs(aux['soft_att'].detach().numpy()[:, 0] - g['soft_att']).max() < 1e-5
    assert np.abs(out.detach().numpy() - g['out']).max() < 2e-5
    loss = O.l1_loss(out, gt)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    loss.backward()
    # reference calls transfer coarse->fine (x1..x16); oracle list is fine->coarse
    for i in range(5):
        w = aux['warp'][4 - i].detach().double()
        st = np.array([w.sum().item(), w.abs().sum().item(), (w * w).sum().item()])
        assert np.allclose(st, g[f'warp{i}_stats'], rtol=1e-5, atol=1e-4), i
    gn = np.array([(p.grad.double().norm().item() if p.grad is not None else 0.0) for p in P.values()])
    assert np.allclose(gn, g['grad_norm'], rtol=2e-3, atol=2e-6)
    assert abs(np.sqrt((gn ** 2).sum()) - float(g['total_grad_norm'])) < 1e-4 * float(g['total_grad_norm'])


def test_three_step_trajectory_matches_reference_step_api(golden_dir):
    g = load(golden_dir, 'trajectory')
    cfg = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    tr = O.OracleTrainer(O.synth_params(cfg, seed=3), cfg)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1234 + 3)
    periods, rw, em = [30, 70], [1, 1], [3e-4, 1e-6]
    for it in range(1, 4):
        t = it - 1                                   # scheduler.step() only when iter>1
        lr = O.cosine_restart_cyclic_lr(t, 2e-4, periods, rw, em)
        rlr = O.cosine_restart_cyclic_lr(t, 1e-4, periods, rw, em)
        assert abs(lr - g['lrs'][it - 1][0]) < 1e-12 and abs(rlr - g['lrs'][it - 1][1]) < 1e-12
        tr.set_lrs(lr, rlr)
        loss, _, out = tr.step(lq, gt, ref)
        assert abs(loss - g['losses'][it - 1]) < 2e-6, (it, loss, g['losses'][it - 1])
    psum = np.array([p.detach().double().sum().item() for p in tr.P.values()])
    assert np.allclose(psum, g['psum'], rtol=0, atol=5e-4)
    assert np.abs(out.numpy() - g['final_out']).max() < 5e-5
    for i, it in enumerate(range(4, 101)):
        lr = O.cosine_restart_cyclic_lr(it - 1, 2e-4, periods, rw, em)
        assert abs(lr - g['lr_table_from_iter4'][i][0]) < 1e-12


def test_psnr_known_answers(golden_dir):
    g = load(golden_dir, 'psnr')
    a, b = T(g['a']), T(g['b'])
    assert abs(O.psnr(O.tensor_to_uint8_img(a), O.tensor_to_uint8_img(b)) - float(g['psnr_u8'])) < 1e-9
    fa = a.numpy().transpose(1, 2, 0); fb = np.clip(b.numpy(), 0, 1).transpose(1, 2, 0)
    assert abs(O.psnr(fa, fb, crop_border=2) - float(g['psnr_float_crop2'])) < 1e-9

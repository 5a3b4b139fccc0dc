"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/tdr.h
declares; the module surface mirrors the reference's names/order/shapes; host logic
(registry errors, schedulers, PSNR, MASA geometry) matches the oracle / golden vectors."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import nafnet_ref_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'tdr.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(tdr_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from textualdegremoval_amd import _lib
    lib = _lib.load()                       # raises if a declared symbol is missing
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/tdr.h but not exported'
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    assert lib.tdr_version() >= 100
    assert lib.tdr_conv_ck(1) == 32 and lib.tdr_conv_ck(3) == 8 and lib.tdr_conv_ck(2) == 16
    assert lib.tdr_packed_weight_floats(40, 3, 3) == 1 * 9 * 8 * 64      # Mpad=64, one chunk of 8 channels


def test_error_convention_no_throw_and_message():
    from textualdegremoval_amd import _lib
    lib = _lib.load()
    rc = lib.tdr_pack_weights(None, 8, 8, 3, 0, None, None)          # null pointers -> error code, no crash
    assert rc < 0 and b'null' in lib.tdr_last_error()
    with pytest.raises(_lib.TdrError):
        _lib.check(rc, 'tdr_pack_weights')


def test_module_surface_matches_reference_registration_order():
    from textualdegremoval_amd.models.archs import define_network
    for kw in (dict(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]),
               dict(width=16, nf=16, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])):
        net = define_network(dict(type='NAFNetRefFusion', enc_blk_nums=[1, 1, 1, 1], dec_blk_nums=[1, 1, 1, 1], **kw))
        sd = net.state_dict()
        shapes = O.param_shapes(O.default_cfg(**kw))
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
        # default init: beta/gamma zeros, LN weight ones / bias zeros (reference :213-214, nafnet_arch_utils.py:295-296)
        assert sd['encoders.0.0.beta'].abs().sum() == 0 and sd['encoders.0.0.norm1.weight'].eq(1).all()
    masa = [k for k in sd if 'masa' in k]
    assert masa == list(sd.keys())[:len(masa)]          # the ref_lr group is a prefix of the registration order


def test_registry_and_step_api_error_behaviour():
    from textualdegremoval_amd.models import create_model
    from textualdegremoval_amd.models.archs import define_network
    with pytest.raises(ValueError):
        define_network({'type': 'NoSuchNet'})
    with pytest.raises(ValueError):
        create_model({'model_type': 'NoSuchModel'})
    with pytest.raises(ValueError):                      # nf != width (concat widths)
        define_network(dict(type='NAFNetRefFusion', width=8, nf=16, enc_blk_nums=[1] * 4, dec_blk_nums=[1] * 4,
                            reffusion_n_blocks=[1] * 5))
    r7 = str(np.load(os.path.join(GOLDEN, 'reference_defects.npz'))['r7_nafnetlocal_reffusion'])
    assert r7.startswith('TypeError')                    # what the reference does (make_golden_defects.py) ...
    with pytest.raises(TypeError, match='ref'):          # ... and the same here: TLSC wrapper of the guided NAFNet, defect R7
        define_network(dict(type='NAFNetLocal_RefFusion', width=8, nf=8, enc_blk_nums=[1] * 4, dec_blk_nums=[1] * 4,
                            reffusion_n_blocks=[1] * 5))
    r8 = [str(v) for v in np.load(os.path.join(GOLDEN, 'reference_defects.npz'))['r8_sfnet_reffusion']]
    assert all(v.startswith('RuntimeError') for v in r8)         # the reference's SFNet-ref never completes a forward (R8)
    sf = define_network(dict(type='SFNetRefFusion', mode='train', num_res=2, nf=32))
    with pytest.raises(RuntimeError, match='R8'):
        sf(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))
    with pytest.raises(IndexError):                      # reference quirk R2: needs len(enc)+1 fusion counts
        define_network(dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=[1] * 4, dec_blk_nums=[1] * 4,
                            reffusion_n_blocks=[1] * 4))
    opt = {'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 0, 'dist': False, 'is_train': True,
           'network_g': dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=[1] * 4, dec_blk_nums=[1] * 4,
                             ext_n_blocks=[1] * 4, reffusion_n_blocks=[1] * 5),
           'path': {}, 'train': {}, 'logger': {}, 'val': {}}
    with pytest.raises(ValueError, match='pixel loss are None'):
        create_model(opt)
    opt['train'] = {'pixel_opt': {'type': 'L1Loss'}, 'optim_g': {'type': 'SGD', 'lr': 1e-3, 'ref_lr': 1e-3}}
    with pytest.raises(NotImplementedError):
        create_model(opt)


def test_no_cpu_fallback_on_the_product_path():
    from textualdegremoval_amd.models.archs import define_network
    net = define_network(dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=[1] * 4, dec_blk_nums=[1] * 4,
                              ext_n_blocks=[1] * 4, reffusion_n_blocks=[1] * 5))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(torch.zeros(1, 3, 128, 128), torch.zeros(1, 3, 128, 128))


def test_scheduler_matches_reference_lr_table():
    from textualdegremoval_amd.models.lr_scheduler import CosineAnnealingRestartCyclicLR
    g = np.load(os.path.join(GOLDEN, 'trajectory.npz'))
    p1, p2 = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{'params': [p1], 'lr': 2e-4}, {'params': [p2], 'lr': 1e-4}], lr=2e-4)
    sch = CosineAnnealingRestartCyclicLR(opt, periods=[30, 70], restart_weights=[1, 1], eta_mins=[3e-4, 1e-6])
    lrs = [[g_['lr'] for g_ in opt.param_groups]]
    for it in range(2, 101):
        opt.step(); sch.step()
        lrs.append([g_['lr'] for g_ in opt.param_groups])
    assert np.allclose(np.array(lrs[:3]), g['lrs'], rtol=0, atol=1e-15)
    assert np.allclose(np.array(lrs[3:]), g['lr_table_from_iter4'], rtol=0, atol=1e-15)


def test_psnr_and_l1_host_semantics():
    from textualdegremoval_amd.losses import L1Loss
    from textualdegremoval_amd.metrics import calculate_psnr, tensor2img
    g = np.load(os.path.join(GOLDEN, 'psnr.npz'))
    a, b = torch.from_numpy(g['a']), torch.from_numpy(g['b'])
    assert abs(calculate_psnr(tensor2img(a, rgb2bgr=False), tensor2img(b, rgb2bgr=False), 0) - float(g['psnr_u8'])) < 1e-9
    assert abs(calculate_psnr(a, b.clamp(0, 1), 2) - float(g['psnr_float_crop2'])) < 1e-9
    with pytest.raises(ValueError):
        L1Loss(reduction='bogus')
    assert abs(L1Loss(0.5)(a, b).item() - 0.5 * (a - b).abs().mean().item()) < 1e-7


def test_masa_geometry_matches_reference_formulas():
    from textualdegremoval_amd.engine import MasaGeom
    g = MasaGeom(512, 512, 512, 512, 4, 8, 1.5, [1, 2, 3])
    assert (g.py, g.px, g.K, g.dia_x, g.side, g.P) == (4, 4, 8, 13, 15, 16)
    g = MasaGeom(128, 128, 128, 128, 4, 8, 1.5, [1, 2, 3])
    assert (g.py, g.px, g.K, g.dia_x, g.side) == (1, 1, 8, 13, 15)                    # cfg1 (wrap case)
    with pytest.raises(ValueError):
        MasaGeom(128, 256, 128, 128, 4, 8, 1.5, [1, 2, 3])                            # non-square: reference crashes too


def test_optimizer_param_indices_follow_the_reference_for_unused_tensors():
    """PromptIR-ref registers six convolutions it never uses.  The reference's setup_optimizers (image_restoration_ref_model.py:
    160-170) puts EVERY named parameter into the two groups (their .grad stays None, AdamW skips them), so the parameter indices
    inside optimizer.state_dict() -- what a `.state` resume file is keyed by -- count them.  Same here."""
    import bench
    from textualdegremoval_amd.models import create_model
    opt = bench.make_opt(32, [1, 1, 1, 1], 64, False, arch='promptir')
    opt['num_gpu'] = 0
    m = create_model(opt)
    named = list(m.net_g.named_parameters())
    groups = m.optimizer_g.param_groups
    assert len(groups) == 2
    assert [id(p) for p in groups[0]['params']] == [id(p) for k, p in named if 'masa' not in k]
    assert [id(p) for p in groups[1]['params']] == [id(p) for k, p in named if 'masa' in k]
    unused = [k for k, _ in named if k.startswith(m.net_g.unused_parameter_prefixes)]
    assert len(unused) >= 6
    sd = m.optimizer_g.state_dict()
    assert sum(len(g['params']) for g in sd['param_groups']) == len(named)


def test_dist_validation_runs_on_local_rank_zero_only(monkeypatch):
    """reference image_restoration_ref_model.py:319-323: LOCAL_RANK 0 validates, every other rank returns 0."""
    import bench
    from textualdegremoval_amd.models import create_model
    opt = bench.make_opt(8, [1, 1, 1, 1], 64, False)
    opt['num_gpu'] = 0
    m = create_model(opt)
    calls = []
    monkeypatch.setattr(m, 'nondist_validation', lambda *a: calls.append(a) or 31.5)
    m.opt['dist'] = True
    monkeypatch.setenv('LOCAL_RANK', '1')
    assert m.validation('loader', 10, None, False, True, True) == 0.
    monkeypatch.setenv('LOCAL_RANK', '0')
    assert m.validation('loader', 10, None, False, True, True) == 31.5 and len(calls) == 1
    m.opt['dist'] = False
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert m.validation('loader', 10, None) == 31.5 and len(calls) == 2


def test_host_side_plane_splits_lose_nothing():
    """kernels.split_planes3 / split_planes (the frozen weights of the token-major GEMMs, split once on the host): three bf16 planes
    sum to the fp32 value bit for bit on any exponent -- the same statement tests/test_hip_dino.py makes for the device-side producers
    (tdr_split3_bf16) -- and the 2-way fp16 split carries 22 significand bits inside the fp16 window."""
    import torch
    from textualdegremoval_amd import kernels as K
    g = torch.Generator().manual_seed(2)
    w = torch.randn(64, 96, generator=g) * torch.logspace(-30, 30, 96)
    p3 = K.split_planes3(w)
    assert p3.shape == (3, 64, 96) and p3.dtype == torch.bfloat16 and p3.is_contiguous()
    assert torch.equal((p3[0].float() + p3[1].float()) + p3[2].float(), w)
    assert (p3[1].float().abs() <= p3[0].float().abs() * 2.0 ** -8 + 1e-45).all()            # each plane is a residual of the one before
    v = torch.randn(32, 40, generator=g)
    p2 = K.split_planes(v)
    assert p2.shape == (2, 32, 40) and p2.dtype == torch.float16
    assert ((p2[0].float() + p2[1].float()) - v).abs().max().item() <= 2.0 ** -21 * v.abs().max().item()

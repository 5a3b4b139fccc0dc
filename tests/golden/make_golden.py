"""Generate golden vectors by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference, which never travels):
    python tests/golden/make_golden.py
Writes tests/golden/*.npz (data only: inputs are regenerated from seeds by
oracle.nafnet_ref_oracle.synth_pair/synth_params, outputs are stored).

Reference defects worked around here:
  R2: reffusion_n_blocks needs len(enc)+1 entries.
  R3: 'fix_iterations' key absent -> all params trainable.
Nothing from the reference is copied: it is imported, executed, and only its
numeric outputs are saved.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import nafnet_ref_oracle as O  # noqa: E402


def import_ref_arch():
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.network_nafnet_guided_arch')


def sample(t, n=32):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def build_ref_net(arch, cfg, P):
    net = arch.NAFNetRefFusion(
        img_channel=cfg['img_channel'], width=cfg['width'], middle_blk_num=cfg['middle_blk_num'],
        enc_blk_nums=cfg['enc_blk_nums'], dec_blk_nums=cfg['dec_blk_nums'], nf=cfg['nf'],
        ext_n_blocks=cfg['ext_n_blocks'], reffusion_n_blocks=cfg['reffusion_n_blocks'],
        lr_block_size=cfg['lr_block_size'], ref_down_block_size=cfg['ref_down_block_size'],
        dilations=cfg['dilations'], psize=cfg['psize'])
    sd = net.state_dict()
    assert list(sd.keys()) == list(P.keys()), 'registration order mismatch'
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    net.load_state_dict(P)
    return net


def whole_net_case(arch, name, cfg, B, H, W, seed, ref_hw=None):
    P = O.synth_params(cfg, seed=seed)
    net = build_ref_net(arch, cfg, P)
    lq, gt, ref = O.synth_pair(B, H, W, seed=1234 + seed, ref_hw=ref_hw)
    rec = {}
    orig_search, orig_search_org, orig_transfer = net.search, net.search_org, net.transfer

    def search(*a, **k):
        r = orig_search(*a, **k); rec['corr_sum_top'] = r[0].detach(); rec['index'] = r[1].detach(); return r

    def search_org(*a, **k):
        r = orig_search_org(*a, **k); rec['soft'] = r[0].detach(); rec['index_all'] = r[1].detach(); return r
    warps = []

    def transfer(*a, **k):
        r = orig_transfer(*a, **k); warps.append(r); return r
    net.search, net.search_org, net.transfer = search, search_org, transfer
    out = net(lq, ref)
    loss = (out - gt).abs().mean()
    loss.backward()
    # top-1 / top-2 gap of the fine search (SURVEY hard part 2)
    d = dict(out=out.detach().numpy(), loss=np.float64(loss.item()),
             index=rec['index'].numpy(), index_all=rec['index_all'][..., 0].numpy(),
             soft_att=rec['soft'][..., 0].numpy(),
             cfg_B=B, cfg_H=H, cfg_W=W, seed=seed)
    for i, wv in enumerate(warps):           # order x1,x2,x4,x8,x16 (coarse->fine)
        d[f'warp{i}_stats'] = stats(wv)
        d[f'warp{i}_sample'] = sample(wv, 64)
    # top-1/top-2 gaps of both arg-max searches (SURVEY hard part 2), from the pinned oracle on the same inputs
    with torch.no_grad():
        _, aux = O.nafnet_ref_forward(P, cfg, lq, ref, return_aux=True)
    assert torch.equal(aux['index_all'], rec['index_all'][..., 0])
    t2 = aux['corr_fine'].topk(2, dim=2).values
    d['fine_gap'] = (t2[..., 0] - t2[..., 1]).numpy()
    c2 = aux['corr_sum'].topk(2, dim=2).values
    d['coarse_gap'] = (c2[..., 0] - c2[..., 1]).numpy()
    names = list(P.keys())
    gnorm = np.zeros(len(names)); gsum = np.zeros(len(names)); gsample = np.zeros((len(names), 8), dtype=np.float32)
    for i, (k, p) in enumerate(net.named_parameters()):
        assert k == names[i]
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        gnorm[i] = g.double().norm().item(); gsum[i] = g.double().sum().item()
        s = sample(g, 8); gsample[i, :len(s)] = s
    d['grad_norm'] = gnorm; d['grad_sum'] = gsum; d['grad_sample'] = gsample
    d['total_grad_norm'] = np.float64(np.sqrt((gnorm ** 2).sum()))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print(name, 'loss', loss.item(), 'gnorm', d['total_grad_norm'], 'idx', rec['index'].flatten()[:8].tolist(),
          'min fine gap', d['fine_gap'].min(), 'min coarse gap', d['coarse_gap'].min())
    return net, P, (lq, gt, ref)


def per_op_cases(arch):
    utils = importlib.import_module('models.archs.nafnet_arch_utils')
    d = {}
    g = torch.Generator().manual_seed(7)
    # a1 LayerNorm2d fwd + custom bwd
    x = torch.randn(2, 12, 9, 10, generator=g, requires_grad=True)
    ln = utils.LayerNorm2d(12)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(12, generator=g) * 0.3 + 1); ln.bias.copy_(torch.randn(12, generator=g) * 0.2)
    go = torch.randn(2, 12, 9, 10, generator=g)
    y = ln(x); y.backward(go)
    d.update(ln_x=x.detach().numpy(), ln_w=ln.weight.detach().numpy(), ln_b=ln.bias.detach().numpy(), ln_go=go.numpy(),
             ln_y=y.detach().numpy(), ln_gx=x.grad.numpy(), ln_gw=ln.weight.grad.numpy(), ln_gb=ln.bias.grad.numpy())
    # a3 NAFBlock fwd/bwd, c=16 @ 16x20
    c = 16
    blk = arch.NAFBlock(c)
    names = [k for k, _ in blk.named_parameters()]
    with torch.no_grad():
        for i, (k, p) in enumerate(blk.named_parameters()):
            gg = torch.Generator().manual_seed(100 + i)
            if k in ('beta', 'gamma'):
                p.copy_(torch.randn(p.shape, generator=gg) * 0.5)
            elif 'norm' in k:
                p.copy_(torch.randn(p.shape, generator=gg) * 0.2 + (1.0 if k.endswith('weight') else 0.0))
            else:
                p.copy_(torch.randn(p.shape, generator=gg) * 0.2)
    x = torch.randn(2, c, 16, 20, generator=g, requires_grad=True)
    go = torch.randn(2, c, 16, 20, generator=g)
    y = blk(x); y.backward(go)
    d.update(naf_x=x.detach().numpy(), naf_go=go.numpy(), naf_y=y.detach().numpy(), naf_gx=x.grad.numpy())
    for k, p in blk.named_parameters():
        d['naf_p_' + k] = p.detach().numpy(); d['naf_g_' + k] = p.grad.numpy()
    d['naf_names'] = np.array(names)
    # a5 ResidualBlock + Encoder (nf=4) fwd/bwd
    enc = arch.Encoder(in_chl=3, nf=4, n_blks=[1, 1, 1, 1])
    with torch.no_grad():
        for i, (k, p) in enumerate(enc.named_parameters()):
            gg = torch.Generator().manual_seed(300 + i)
            p.copy_(torch.randn(p.shape, generator=gg) * (0.25 if k.endswith('weight') else 0.1))
    x = torch.rand(1, 3, 32, 48, generator=g, requires_grad=True)
    feats = enc(x)
    loss = sum((f * f).mean() * (i + 1) for i, f in enumerate(feats))
    loss.backward()
    d['enc_x'] = x.detach().numpy(); d['enc_gx'] = x.grad.numpy()
    d['enc_names'] = np.array([k for k, _ in enc.named_parameters()])
    for i, f in enumerate(feats):
        d[f'enc_f{i}'] = f.detach().numpy()
    for k, p in enc.named_parameters():
        d['enc_p_' + k] = p.detach().numpy(); d['enc_g_' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'per_op.npz'), **d)
    print('per_op done')


def masa_ops_case(arch):
    """search / search_org / transfer in isolation incl. gradients, C=8."""
    net = arch.NAFNetRefFusion(width=8, enc_blk_nums=[1, 1, 1, 1], dec_blk_nums=[1, 1, 1, 1], nf=8,
                               ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    g = torch.Generator().manual_seed(11)
    d = {}
    lr = torch.randn(3, 8, 10, 10, generator=g, requires_grad=True)
    rf = torch.randn(3, 8, 15, 15, generator=g, requires_grad=True)
    corr, idx = net.search_org(lr, rf, ks=3, pd=1, stride=1)
    go = torch.randn(corr.shape, generator=g)
    (corr * go).sum().backward()
    d.update(so_lr=lr.detach().numpy(), so_ref=rf.detach().numpy(), so_val=corr.detach().numpy()[..., 0],
             so_idx=idx.numpy()[..., 0], so_go=go.numpy()[..., 0], so_glr=lr.grad.numpy(), so_gref=rf.grad.numpy())
    for s in (1, 2, 4):
        fea = torch.randn(3, 4, 15 * s, 15 * s, generator=g, requires_grad=True)
        att = torch.rand(3, 1, 8, 8, generator=g, requires_grad=True)
        index = torch.randint(0, 169, (3, 8, 8), generator=g)
        o = net.transfer(fea, index, att, ks=3 * s, pd=s, stride=s)
        go = torch.randn(o.shape, generator=g)
        (o * go).sum().backward()
        d.update({f'tr{s}_fea': fea.detach().numpy(), f'tr{s}_att': att.detach().numpy(), f'tr{s}_idx': index.numpy(),
                  f'tr{s}_out': o.detach().numpy(), f'tr{s}_go': go.numpy(), f'tr{s}_gfea': fea.grad.numpy(),
                  f'tr{s}_gatt': att.grad.numpy()})
    lrp = torch.randn(2, 4, 8, 10, 10, generator=g)
    reff = torch.randn(2, 8, 12, 12, generator=g)
    sc, ind = net.search(lrp, reff, ks=3, pd=1, stride=1, dilations=[1, 2, 3])
    full = 0
    d.update(cs_lr=lrp.numpy(), cs_ref=reff.numpy(), cs_idx=ind.numpy()[..., 0], cs_val=sc.numpy()[..., 0])
    np.savez_compressed(os.path.join(HERE, 'masa_ops.npz'), **d)
    print('masa_ops done')


def trajectory_case():
    """3 steps of the reference RefGuidedImageCleanModel.optimize_parameters on
    CPU (full step API, SURVEY 8c recipe 2): loss, LRs, parameter checksums."""
    for k in ('models', 'models.archs'):
        sys.modules.pop(k, None)
    for k in [k for k in sys.modules if k.startswith('models.')]:
        sys.modules.pop(k)
    import transformers  # noqa: F401  (must precede the stubs)
    for name in ('cv2', 'torchvision', 'torchvision.utils', 'skimage', 'skimage.metrics', 'lmdb'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision.utils'].make_grid = lambda *a, **k: None
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    sys.modules['skimage'].metrics = sys.modules['skimage.metrics']
    import tempfile
    from models import create_model
    from models.dino.vision_transformers import vit_base
    cfg = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    P = O.synth_params(cfg, seed=3)
    tmp = tempfile.mkdtemp()
    torch.manual_seed(0)
    dino = vit_base(img_size=518, patch_size=14, init_values=1.0, ffn_layer='mlp', block_chunks=0)
    torch.save(dino.state_dict(), tmp + '/dino.pth')
    torch.save({'params': P}, tmp + '/net.pth')
    opt = {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 0, 'dist': False, 'is_train': True, 'rank': 0,
        'world_size': 1,
        'network_g': dict(type='NAFNetRefFusion', width=8, nf=8, enc_blk_nums=[1, 1, 1, 1], dec_blk_nums=[1, 1, 1, 1],
                          middle_blk_num=1, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1]),
        'path': {'pretrain_dino': tmp + '/dino.pth', 'pretrain_network_g': tmp + '/net.pth', 'strict_load_g': True,
                 'param_key': 'params'},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [30, 70], 'restart_weights': [1, 1],
                                'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                  'use_grad_clip': True, 'total_iter': 100, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'val': {}, 'scale': 1,
    }
    model = create_model(opt)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1234 + 3)
    losses, lrs, outs = [], [], []
    for it in range(1, 4):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data({'lq': lq, 'gt': gt, 'ref': ref})
        model.optimize_parameters(it)
        losses.append(model.get_current_log()['l_pix'])
        lrs.append(model.get_current_learning_rate())
        assert torch.equal(model.ref_in, ref)
    sd = model.net_g.state_dict()
    names = list(sd.keys())
    psum = np.array([sd[k].double().sum().item() for k in names])
    pabs = np.array([sd[k].double().abs().sum().item() for k in names])
    d = dict(losses=np.array(losses), lrs=np.array(lrs), psum=psum, pabs=pabs,
             p_sample=np.stack([np.pad(sample(sd[k], 4), (0, 4 - min(4, sd[k].numel()))) for k in names]),
             final_out=model.output.detach().numpy())
    # scheduler table over 100 iters
    table = []
    sched = model.schedulers[0]
    for it in range(4, 101):
        model.update_learning_rate(it, warmup_iter=-1)
        table.append(model.get_current_learning_rate())
    d['lr_table_from_iter4'] = np.array(table)
    np.savez_compressed(os.path.join(HERE, 'trajectory.npz'), **d)
    print('trajectory', losses, lrs)


def psnr_case():
    rng = np.random.RandomState(0)
    a = rng.rand(3, 24, 20).astype(np.float32)
    b = np.clip(a + rng.randn(3, 24, 20).astype(np.float32) * 0.05, -0.1, 1.1)
    # metrics/psnr_ssim.py needs cv2/skimage -> stubbed modules already present
    from metrics.psnr_ssim import calculate_psnr
    ia = np.round(np.clip(a, 0, 1).transpose(1, 2, 0) * 255.0).astype(np.uint8)
    ib = np.round(np.clip(b, 0, 1).transpose(1, 2, 0) * 255.0).astype(np.uint8)
    v8 = calculate_psnr(ia, ib, crop_border=0)
    vf = calculate_psnr(torch.from_numpy(a), torch.from_numpy(np.clip(b, 0, 1)), crop_border=2)
    np.savez_compressed(os.path.join(HERE, 'psnr.npz'), a=a, b=b, psnr_u8=v8, psnr_float_crop2=vf)
    print('psnr', v8, vf)


if __name__ == '__main__':
    torch.set_num_threads(8)
    arch = import_ref_arch()
    only_net = len(sys.argv) > 1 and sys.argv[1] in ('net', 'refsize', 'yaml64')
    if len(sys.argv) > 1 and sys.argv[1] == 'yaml64':
        # the widths of the one shipped NAFNet YAML (002_nafnet_single_image_motion_deblurring.yml:45-61: width = nf = 64,
        # enc [1,1,1,28], dec [1,1,1,1], ext [4,4,4,4]); the deep stack cut to 3 blocks, and reffusion_n_blocks given the five
        # entries the class needs (the YAML's four raise IndexError, quirk R2).  Channels reach 2048 in the middle fusion block.
        y64 = O.default_cfg(width=64, nf=64, enc_blk_nums=[1, 1, 1, 3], dec_blk_nums=[1, 1, 1, 1], middle_blk_num=1,
                            ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 1])
        whole_net_case(arch, 'net_yaml_w64_128', y64, 1, 128, 128, seed=12)
        sys.exit(0)
    small = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    # ref of another size than lq (validation / inference: the full generated reference against any lq,
    # image_restoration_ref_model.py:286-330): block diameter follows the ref size (:606-607)
    if len(sys.argv) < 2 or sys.argv[1] == 'refsize':
        whole_net_case(arch, 'net_w8_256_ref384', small, 1, 256, 256, seed=8, ref_hw=(384, 384))      # 21x21 boxes in a 24x24 map
        whole_net_case(arch, 'net_w8_128_ref256_wrap', small, 1, 128, 128, seed=9, ref_hw=(256, 256))  # 27x27 boxes wrap a 16x16 map
        whole_net_case(arch, 'net_w8_200x136_ref300', small, 1, 200, 136, seed=10, ref_hw=(300, 300))  # zero-padded to 256x256 / 384x384
        if len(sys.argv) > 1:
            sys.exit(0)
    if not only_net:
        per_op_cases(arch)
        masa_ops_case(arch)
    small = O.default_cfg(width=8, nf=8, ext_n_blocks=[1, 1, 1, 1], reffusion_n_blocks=[1, 1, 1, 1, 1])
    whole_net_case(arch, 'net_w8_128_wrap', small, 1, 128, 128, seed=1)
    whole_net_case(arch, 'net_w8_256_b2', small, 2, 256, 256, seed=2)       # contains a near-tie (gap < 1e-6)
    whole_net_case(arch, 'net_w8_256_b2_clear', small, 2, 256, 256, seed=6)
    whole_net_case(arch, 'net_w8_120x100_pad', small, 1, 120, 100, seed=4)
    cfg1 = O.default_cfg(width=16, nf=16, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])
    whole_net_case(arch, 'net_cfg1_w16_128', cfg1, 1, 128, 128, seed=5)
    if not only_net:
        trajectory_case()
        psnr_case()

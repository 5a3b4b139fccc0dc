"""Generate golden vectors for the validation-path pieces of SURVEY 8 row f2 by running the REFERENCE on CPU.

Run in the build container only:   python tests/golden/make_golden_tlsc.py
Writes tests/golden/tlsc.npz (data only):
  * `AvgPool2d` of models/archs/nafnet_local_arch.py:10-75 (fast_imp False) on a few shapes / kernel sizes, incl. kernels larger
    than one or both sides;
  * `NAFNetLocal` of models/archs/network_nafnet_guided_arch.py:756-768 (TLSC wrapper of the un-guided NAFNet: constructed with
    a small train_size, run on a larger image so that every level pools locally), weights stored;
  * `_ssim_cly` of metrics/psnr_ssim.py:184-222, located with `ast` and executed from the reference file.  Its module imports cv2
    (absent here) for two calls: cv2.getGaussianKernel(11, 1.5) and cv2.filter2D(img, -1, window, borderType=BORDER_REPLICATE).
    They are provided by a namespace object backed by scipy.ndimage.correlate(mode='nearest') (an independent implementation
    of correlation with replicated borders) and the kernel's published formula -- the formula of the reference runs unchanged."""
import ast
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def import_ref(name):
    sys.path.insert(0, REF)
    m = types.ModuleType('models'); m.__path__ = [REF + '/models']; sys.modules['models'] = m
    a = types.ModuleType('models.archs'); a.__path__ = [REF + '/models/archs']; sys.modules['models.archs'] = a
    return importlib.import_module('models.archs.' + name)


def pool_cases(d):
    loc = import_ref('nafnet_local_arch')
    g = torch.Generator().manual_seed(1)
    for i, (shape, k) in enumerate([((2, 3, 20, 28), (7, 9)), ((1, 4, 33, 17), (12, 30)), ((1, 2, 16, 40), (16, 11)), ((1, 5, 24, 24), (5, 5)),
                                    ((2, 2, 9, 64), (4, 48))]):
        x = torch.randn(shape, generator=g)
        p = loc.AvgPool2d(kernel_size=list(k), auto_pad=True, fast_imp=False)
        d[f'pool{i}_x'], d[f'pool{i}_k'], d[f'pool{i}_out'] = x.numpy(), np.array(k), p(x).numpy()
    d['pool_n'] = np.array(5)


def naflocal_case(d):
    naf = import_ref('network_nafnet_guided_arch')
    torch.manual_seed(7)
    net = naf.NAFNetLocal(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1], dec_blk_nums=[1, 1], train_size=(1, 3, 32, 32))
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.1 if p.dim() <= 1 or k.endswith(('beta', 'gamma')) else 0)
    ks = [tuple(m.kernel_size) for m in net.modules() if isinstance(m, sys.modules['models.archs.nafnet_local_arch'].AvgPool2d)]
    x = torch.rand(1, 3, 78, 94, generator=g)                  # padded to 80 x 96 (levels 80x96, 40x48, 20x24: rows of 4-pixel multiples, the depthwise stencil's requirement); every level is larger than its pooling kernel
    with torch.no_grad():
        out = net(x)
    d['nl_x'], d['nl_out'], d['nl_ksizes'] = x.numpy(), out.numpy(), np.array(ks)
    d['nl_names'] = np.array([k for k, _ in net.named_parameters()])
    for k, p in net.named_parameters():
        d['nl_p_' + k] = p.detach().numpy()
    print('NAFNetLocal', tuple(out.shape), float(out.abs().mean()), ks)


def ssim_cases(d):
    from scipy import ndimage
    src = open(REF + '/metrics/psnr_ssim.py').read()
    node = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == '_ssim_cly'][0]

    def gk(ksize, sigma):
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(x * x) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(-1, 1)
    cv2 = types.SimpleNamespace(getGaussianKernel=gk, BORDER_REPLICATE=1,
                                filter2D=lambda img, ddepth, window, borderType: ndimage.correlate(img, window, mode='nearest'))
    ns = {'np': np, 'cv2': cv2}
    exec(compile(ast.Module([node], []), REF + '/metrics/psnr_ssim.py', 'exec'), ns)
    rng = np.random.default_rng(5)
    n = 0
    for (h, w, noise) in [(40, 36, 12.0), (17, 64, 3.0), (9, 11, 30.0), (64, 64, 0.5)]:
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 90 * np.sin(yy / 7.0) * np.cos(xx / 5.0) + rng.normal(0, 20, (h, w))
        a = np.clip(base, 16, 235).astype(np.float32)
        b = np.clip(base + rng.normal(0, noise, (h, w)), 16, 235).astype(np.float32)
        d[f'ssim{n}_a'], d[f'ssim{n}_b'], d[f'ssim{n}_val'] = a, b, np.array(float(ns['_ssim_cly'](a, b)))
        print('ssim_cly', (h, w), float(d[f'ssim{n}_val']))
        n += 1
    d['ssim_n'] = np.array(n)


if __name__ == '__main__':
    d = {}
    pool_cases(d)
    naflocal_case(d)
    ssim_cases(d)
    np.savez_compressed(os.path.join(HERE, 'tlsc.npz'), **d)
    print('wrote tlsc.npz', len(d), 'arrays')

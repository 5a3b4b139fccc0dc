"""Generate golden vectors for the host-side input pipeline by RUNNING the reference:

  * `paired_random_crop` and `random_augmentation` / `data_augmentation` of /root/reference/data/transforms.py
    (:24-84, :223-275), imported as a module (its only missing dependency, cv2, is stubbed -- neither function touches it);
  * the sigma-noise synthesis of Dataset_GaussianDenoisingWithRef.__getitem__ (data/restoration_dataset.py:464-476): those
    statements are read from the reference file by line range and executed here on a stub `self` (the class itself needs
    file clients, cv2 decoding and image folders); nothing of the reference is copied into the repo, only the numeric
    outputs are stored.

Run in the build container only:   python tests/golden/make_golden_transforms.py
Writes tests/golden/transforms.npz.  Inputs are regenerated from seeds by the tests.

`padding()` (utils/utils_image.py:243-259) is cv2.copyMakeBorder(BORDER_REFLECT) and cannot run here (cv2 absent): the
oracle restates it as numpy's 'symmetric' pad ("fedcba|abcdefgh|hgfedcb" in OpenCV's border table) -- that one sub-step
stays "parity unpinned" and says so in oracle/data_pipeline_oracle.py."""
import os
import random
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def load_transforms():
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'data' or k.startswith('data.')]:
        sys.modules.pop(k)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_transforms', os.path.join(REF, 'data', 'transforms.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def image(seed, h, w):
    return np.random.RandomState(seed).rand(h, w, 3).astype(np.float32)


def main():
    T = load_transforms()
    d = {}
    # (a) every augmentation mode on a non-square image, straight from data_augmentation
    img = image(1, 5, 7)
    for mode in range(8):
        d[f'mode{mode}'] = np.ascontiguousarray(T.data_augmentation(img, mode))
    # (b) the per-sample train path: paired_random_crop then random_augmentation, python `random` seeded per case
    cases = [(11, 40, 52, 32), (12, 64, 64, 64), (13, 96, 80, 48), (14, 33, 57, 32), (15, 128, 128, 96), (16, 50, 50, 16)]
    d['cases'] = np.array(cases)
    for seed, h, w, patch in cases:
        gt, lq = image(seed, h, w), image(seed + 1000, h, w)
        random.seed(seed)
        g, q = T.paired_random_crop(gt, lq, patch, 1, 'unused_path')
        g, q = T.random_augmentation(g, q)
        d[f'c{seed}_gt'] = np.ascontiguousarray(g)
        d[f'c{seed}_lq'] = np.ascontiguousarray(q)
        # the draws themselves (replayed: randint(0, h-p), randint(0, w-p), randint(0, 7) -- the order the reference consumed them in)
        random.seed(seed)
        d[f'c{seed}_draws'] = np.array([random.randint(0, h - patch), random.randint(0, w - patch), random.randint(0, 7)])
    # (c) error behaviour: patch larger than the image without padding()
    try:
        T.paired_random_crop(image(2, 20, 20), image(3, 20, 20), 32, 1, 'p')
        d['small_raises'] = np.array(0)
    except ValueError:
        d['small_raises'] = np.array(1)
    # (d) sigma-noise synthesis: the reference's statements, by line range, on a stub self
    src = open(os.path.join(REF, 'data', 'restoration_dataset.py')).read().splitlines()
    first = next(i for i, ln in enumerate(src) if i > 450 and "if self.sigma_type == 'constant':" in ln)
    last = next(i for i in range(first, first + 20) if 'img_lq.add_(noise)' in src[i])
    block = textwrap.dedent('\n'.join(src[first:last + 1]))
    d['noise_lines'] = np.array([first + 1, last + 1])
    for tag, stype, srange in (('const', 'constant', 15), ('rand', 'random', [0, 55]), ('choice', 'choice', [15, 25, 50])):
        self = types.SimpleNamespace(sigma_type=stype, sigma_range=srange)
        img_lq = torch.from_numpy(image(30, 24, 20).transpose(2, 0, 1).copy())
        random.seed(77)
        torch.manual_seed(78)
        ns = {'self': self, 'img_lq': img_lq, 'random': random, 'torch': torch}
        exec(block, ns)
        d[f'noise_{tag}_out'] = ns['img_lq'].numpy()
        d[f'noise_{tag}_sigma'] = np.array(float(ns['sigma_value']))
    np.savez_compressed(os.path.join(HERE, 'transforms.npz'), **d)
    print('transforms.npz:', len(d), 'arrays; noise statements = restoration_dataset.py lines', d['noise_lines'])


if __name__ == '__main__':
    main()

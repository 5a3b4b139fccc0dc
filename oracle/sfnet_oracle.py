"""TEST INFRASTRUCTURE (checker only -- never imported by the product path): functional torch-CPU restatement of the reference's
UN-GUIDED `SFNet` (models/archs/network_sfnet_guided_arch.py:320-407) and of the operators it is built from
(models/archs/sfnet_arch_utils.py:76-265), keyed by the reference's state-dict names.  Training-mode semantics only (mode[0] == 'train':
every pooling is a global average, BatchNorm2d normalises with the statistics of the batch and moves its running buffers).

Pinned by tests/golden/sfnet.npz (tests/golden/make_golden_sfnet.py runs the reference classes): tests/test_sfnet_oracle_golden.py.

The guided class of the same file (`SFNetRefFusion`, :410-797) cannot run a forward pass in the reference (defect R8): only
this un-guided network exists as an executable specification.

Operators (reference line -> function here):
  BasicConv (:76-98)            conv / transposed conv (4x4, stride 2, padding 1) + bias, optional exact-erf GELU -> basic_conv
  Gap (:101-117)                x_d = mean_HW(x); out = x_d * fscale_d + (x - x_d) * (fscale_h + 1)                -> region_affine(q = 1)
  Patch_ap (:239-265)           the same per image QUADRANT (`(p1 w1)` splits the axis into halves, p1 outer) with per-(channel, quadrant)
                                gains: out = (x - low) * h + low * l                                                 -> region_affine(q = 2)
  dynamic_filter (:152-192)     taps = softmax_k2(BN_batch(conv1x1(mean_HW(x)))) per (image, group of c / 8 channels); low = the k x k
                                stencil of the reflection-padded input with those taps; high = x - low; SFconv re-weights the two bands
  SFconv (:195-236)             z = fc(mean_HW(low + high)); [a_high ; a_low] = softmax over ALL 2c entries of [fcs0(z) ; fcs1(z)];
                                out = conv1x1(high * a_high + low * a_low)
  ResBlock (:120-149), EBlock / DBlock (:178-197), SCM (:200-214: four convs + InstanceNorm2d(affine)), FAM (:217-223), SFNet.forward (:366-407)
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BASE = 32
GROUP = 8
BN_EPS, BN_MOM, IN_EPS = 1e-5, 0.1, 1e-5


# ------------------------------------------------------------------------------------------------------------------ parameters
def _res_block_shapes(pre, c, filt, out):
    out[pre + 'conv1.main.0.weight'] = (c, c, 3, 3)
    out[pre + 'conv1.main.0.bias'] = (c,)
    out[pre + 'conv2.main.0.weight'] = (c, c, 3, 3)
    out[pre + 'conv2.main.0.bias'] = (c,)
    if filt:
        h = c // 2
        d = max(h // 2, 32)
        for name, k in (('dyna', 3), ('dyna_2', 5)):
            p = pre + name + '.'
            out[p + 'lamb_l'] = (h,)
            out[p + 'lamb_h'] = (h,)
            out[p + 'conv.weight'] = (GROUP * k * k, h, 1, 1)
            out[p + 'bn.weight'] = (GROUP * k * k,)
            out[p + 'bn.bias'] = (GROUP * k * k,)
            out[p + 'bn.running_mean'] = (GROUP * k * k,)
            out[p + 'bn.running_var'] = (GROUP * k * k,)
            out[p + 'bn.num_batches_tracked'] = ()
            out[p + 'modulate.fc.weight'] = (d, h, 1, 1)
            out[p + 'modulate.fc.bias'] = (d,)
            for i in range(2):
                out[p + f'modulate.fcs.{i}.weight'] = (h, d, 1, 1)
                out[p + f'modulate.fcs.{i}.bias'] = (h,)
            out[p + 'modulate.out.weight'] = (h, h, 1, 1)
            out[p + 'modulate.out.bias'] = (h,)
    out[pre + 'localap.h'] = (c // 2 * 4,)
    out[pre + 'localap.l'] = (c // 2 * 4,)
    out[pre + 'global_ap.fscale_d'] = (c // 2,)
    out[pre + 'global_ap.fscale_h'] = (c // 2,)


def state_shapes(num_res):
    """OrderedDict name -> shape of the reference's state dict, in its registration order (checked against the reference class by the
    golden generator)"""
    o = OrderedDict()
    b = BASE
    for i, c in enumerate((b, 2 * b, 4 * b)):
        for r in range(num_res):
            _res_block_shapes(f'Encoder.{i}.layers.{r}.', c, r == num_res - 1, o)
    for i, (co, ci, k) in enumerate(((b, 3, 3), (2 * b, b, 3), (4 * b, 2 * b, 3))):
        o[f'feat_extract.{i}.main.0.weight'] = (co, ci, k, k)
        o[f'feat_extract.{i}.main.0.bias'] = (co,)
    for i, (ci, co) in ((3, (4 * b, 2 * b)), (4, (2 * b, b))):                 # ConvTranspose2d: weight [Cin, Cout, 4, 4]
        o[f'feat_extract.{i}.main.0.weight'] = (ci, co, 4, 4)
        o[f'feat_extract.{i}.main.0.bias'] = (co,)
    o['feat_extract.5.main.0.weight'] = (3, b, 3, 3)
    o['feat_extract.5.main.0.bias'] = (3,)
    for i, c in enumerate((4 * b, 2 * b, b)):
        for r in range(num_res):
            _res_block_shapes(f'Decoder.{i}.layers.{r}.', c, r == num_res - 1, o)
    for i, (co, ci) in enumerate(((2 * b, 4 * b), (b, 2 * b))):
        o[f'Convs.{i}.main.0.weight'] = (co, ci, 1, 1)
        o[f'Convs.{i}.main.0.bias'] = (co,)
    for i, ci in enumerate((4 * b, 2 * b)):
        o[f'ConvsOut.{i}.main.0.weight'] = (3, ci, 3, 3)
        o[f'ConvsOut.{i}.main.0.bias'] = (3,)
    for tag, c in (('1', 4 * b), ('2', 2 * b)):
        o[f'FAM{tag}.merge.main.0.weight'] = (c, 2 * c, 3, 3)
        o[f'FAM{tag}.merge.main.0.bias'] = (c,)
        for j, (co, ci, k) in enumerate(((c // 4, 3, 3), (c // 2, c // 4, 1), (c // 2, c // 2, 3), (c, c // 2, 1))):
            o[f'SCM{tag}.main.{j}.main.0.weight'] = (co, ci, k, k)
            o[f'SCM{tag}.main.{j}.main.0.bias'] = (co,)
        o[f'SCM{tag}.main.4.weight'] = (c,)
        o[f'SCM{tag}.main.4.bias'] = (c,)
    return o


def is_buffer(name):
    return name.endswith(('running_mean', 'running_var', 'num_batches_tracked'))


def synth_state(num_res, seed=0):
    """seeded state dict with NON-trivial values everywhere (the reference initialises the band gains to zero, which would hide them):
    conv weights ~ N(0, 1 / sqrt(fan_in)), biases and gains ~ N(0, 0.2), norm weights ~ 1 + N(0, 0.2), running_var ~ U(0.5, 1.5)"""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in state_shapes(num_res).items():
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(3, dtype=torch.long)
        elif k.endswith('running_var'):
            sd[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith('running_mean'):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        elif len(shp) == 4:
            fan = shp[1] * shp[2] * shp[3]
            if k.startswith('feat_extract.3.') or k.startswith('feat_extract.4.'):
                fan = shp[0] * 4                                            # transposed conv: 4 of the 16 taps reach an output pixel
            sd[k] = torch.randn(shp, generator=g) / fan ** 0.5
        elif k.endswith(('bn.weight', 'main.4.weight')):
            sd[k] = 1.0 + 0.2 * torch.randn(shp, generator=g)
        else:
            sd[k] = 0.2 * torch.randn(shp, generator=g)
    return sd


# ------------------------------------------------------------------------------------------------------------------ operators
def basic_conv(x, P, pre, k, stride=1, act=True, transpose=False):
    w, b = P[pre + 'main.0.weight'], P[pre + 'main.0.bias']
    if transpose:
        y = F.conv_transpose2d(x, w, b, stride=stride, padding=k // 2 - 1)
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=k // 2)
    return F.gelu(y) if act else y


def region_affine(x, A, B, q):
    """out = x * A[c, region] + mean_region(x) * B[c, region]; regions = the q x q equal blocks of the image (q = 1: the whole image).
    A, B: [C * q * q] in the reference's `(c p1 p2)` order"""
    n, c, h, w = x.shape
    xr = x.view(n, c, q, h // q, q, w // q)
    m = xr.mean((3, 5), keepdim=True)
    Ar, Br = A.view(1, c, q, 1, q, 1), B.view(1, c, q, 1, q, 1)
    return (xr * Ar + m * Br).reshape(n, c, h, w)


TLSC_TRAIN_SIZE = 256          # sfnet_arch_utils.py:8
TLSC_BASE = {'Indoor': 246, 'Outdoor': 210}     # :110-113, :226-229, :247-250


def tlsc_avgpool(x, base):
    """AvgPool2d(base_size=base) of mode[0] == 'test' (sfnet_arch_utils.py:11-70, the exact branch): the mean over a k1 x k2 window,
    k = size * base // 256 taken from the input it runs on, every valid window position, replicate-padded back to the input size.
    (The reference caches k from the FIRST input a module instance sees; one kernel size per call is what a fresh network does.)"""
    n, c, h, w = x.shape
    k1, k2 = min(h, h * base // TLSC_TRAIN_SIZE), min(w, w * base // TLSC_TRAIN_SIZE)
    out = F.avg_pool2d(x, (k1, k2), stride=1)
    _h, _w = out.shape[2:]
    return F.pad(out, ((w - _w) // 2, (w - _w + 1) // 2, (h - _h) // 2, (h - _h + 1) // 2), mode='replicate')


def gap_module(x, P, pre, tlsc=None):
    fd, fh = P[pre + 'fscale_d'], P[pre + 'fscale_h']
    if tlsc is not None:                            # :115-119 with the local box mean
        x_d = tlsc_avgpool(x, tlsc)
        return x_d * fd.view(1, -1, 1, 1) + (x - x_d) * (fh.view(1, -1, 1, 1) + 1.0)
    return region_affine(x, fh + 1.0, fd - fh - 1.0, 1)


def patch_ap(x, P, pre, tlsc=None):
    h, l = P[pre + 'h'], P[pre + 'l']
    if tlsc is not None:                            # :256-265: the box mean runs on the [b, (c p1 p2), H / 2, W / 2] quadrant planes
        n, c, H, W = x.shape
        px = x.view(n, c, 2, H // 2, 2, W // 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c * 4, H // 2, W // 2)
        low = tlsc_avgpool(px, tlsc)
        out = (px - low) * h.view(1, -1, 1, 1) + low * l.view(1, -1, 1, 1)
        return out.view(n, c, 2, 2, H // 2, W // 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, H, W)
    return region_affine(x, h, l - h, 2)


def batch_norm_eval(v, P, pre):
    """BatchNorm2d after module.eval(): the running statistics normalise, nothing moves"""
    mu, var = P[pre + 'running_mean'].view(1, -1, 1, 1), P[pre + 'running_var'].view(1, -1, 1, 1)
    return (v - mu) / torch.sqrt(var + BN_EPS) * P[pre + 'weight'].view(1, -1, 1, 1) + P[pre + 'bias'].view(1, -1, 1, 1)


def batch_norm_train(v, P, pre, new_buffers=None):
    """BatchNorm2d in training mode on [N, C, 1, 1]: batch statistics (biased variance) normalise, the running buffers move with
    momentum 0.1 towards the batch mean and the UNBIASED batch variance"""
    n = v.shape[0]
    mu = v.mean(0, keepdim=True)
    var = ((v - mu) ** 2).mean(0, keepdim=True)
    if new_buffers is not None:
        cnt = v.numel() // v.shape[1]
        new_buffers[pre + 'running_mean'] = (1 - BN_MOM) * P[pre + 'running_mean'] + BN_MOM * mu.detach().flatten()
        new_buffers[pre + 'running_var'] = (1 - BN_MOM) * P[pre + 'running_var'] + BN_MOM * var.detach().flatten() * cnt / max(cnt - 1, 1)
        new_buffers[pre + 'num_batches_tracked'] = P[pre + 'num_batches_tracked'] + 1
    return (v - mu) / torch.sqrt(var + BN_EPS) * P[pre + 'weight'].view(1, -1, 1, 1) + P[pre + 'bias'].view(1, -1, 1, 1)


def dynamic_filter(x, P, pre, k, new_buffers=None, training=True, tlsc=None):
    n, c, h, w = x.shape
    ap = x.mean((2, 3), keepdim=True)
    lf = F.conv2d(ap, P[pre + 'conv.weight'])
    lf = batch_norm_train(lf, P, pre + 'bn.', new_buffers) if training else batch_norm_eval(lf, P, pre + 'bn.')
    taps = torch.softmax(lf.view(n, GROUP, k * k), dim=2)                       # [N, G, k*k]
    xp = F.pad(x, (k // 2,) * 4, mode='reflect')
    low = torch.zeros_like(x)
    cg = c // GROUP
    for t in range(k * k):
        dy, dx = t // k, t % k
        low = low + xp[:, :, dy:dy + h, dx:dx + w] * taps[:, :, t].repeat_interleave(cg, dim=1).view(n, c, 1, 1)
    high = x - low
    # SFconv (modulate)
    emerge = low + high
    emerge = emerge.mean((2, 3), keepdim=True) if tlsc is None else tlsc_avgpool(emerge, tlsc)     # SFconv.gap (:212-218); dynamic_filter.ap above stays global
    z = F.conv2d(emerge, P[pre + 'modulate.fc.weight'], P[pre + 'modulate.fc.bias'])
    a_h = F.conv2d(z, P[pre + 'modulate.fcs.0.weight'], P[pre + 'modulate.fcs.0.bias'])
    a_l = F.conv2d(z, P[pre + 'modulate.fcs.1.weight'], P[pre + 'modulate.fcs.1.bias'])
    att = torch.softmax(torch.cat([a_h, a_l], 1), dim=1)
    a_h, a_l = att[:, :c], att[:, c:]
    return F.conv2d(high * a_h + low * a_l, P[pre + 'modulate.out.weight'], P[pre + 'modulate.out.bias'])


def res_block(x, P, pre, filt, new_buffers=None, training=True, tlsc=None):
    out = basic_conv(x, P, pre + 'conv1.', 3)
    c = out.shape[1]
    if filt:
        out = torch.cat([dynamic_filter(out[:, :c // 2], P, pre + 'dyna.', 3, new_buffers, training, tlsc),
                         dynamic_filter(out[:, c // 2:], P, pre + 'dyna_2.', 5, new_buffers, training, tlsc)], 1)
    out = torch.cat([gap_module(out[:, :c // 2], P, pre + 'global_ap.', tlsc), patch_ap(out[:, c // 2:], P, pre + 'localap.', tlsc)], 1)
    return basic_conv(out, P, pre + 'conv2.', 3, act=False) + x


def blocks(x, P, pre, num_res, new_buffers=None, training=True, tlsc=None):
    for r in range(num_res):
        x = res_block(x, P, f'{pre}layers.{r}.', r == num_res - 1, new_buffers, training, tlsc)
    return x


def scm(x, P, pre):
    x = basic_conv(x, P, pre + 'main.0.', 3)
    x = basic_conv(x, P, pre + 'main.1.', 1)
    x = basic_conv(x, P, pre + 'main.2.', 3)
    x = basic_conv(x, P, pre + 'main.3.', 1, act=False)
    return F.instance_norm(x, weight=P[pre + 'main.4.weight'], bias=P[pre + 'main.4.bias'], eps=IN_EPS)


def fam(x1, x2, P, pre):
    return basic_conv(torch.cat([x1, x2], 1), P, pre + 'merge.', 3, act=False)


def sfnet_forward(P, x, num_res, new_buffers=None, training=True, tlsc=None):
    """-> [out at 1/4, out at 1/2, out at full size] (reference :366-407).  new_buffers: dict that receives the BatchNorm buffers after
    this training-mode forward pass; training=False: the network after .eval() (BatchNorm2d on its running statistics); tlsc = TLSC_BASE[mode[1]]: the
    mode[0] == 'test' network (Gap, Patch_ap and SFconv pool with the local box mean instead of the global average)"""
    x_2 = x[:, :, ::2, ::2]                        # F.interpolate(scale_factor=0.5), mode 'nearest'
    x_4 = x_2[:, :, ::2, ::2]
    z2 = scm(x_2, P, 'SCM2.')
    z4 = scm(x_4, P, 'SCM1.')
    x_ = basic_conv(x, P, 'feat_extract.0.', 3)
    res1 = blocks(x_, P, 'Encoder.0.', num_res, new_buffers, training, tlsc)
    z = basic_conv(res1, P, 'feat_extract.1.', 3, stride=2)
    z = fam(z, z2, P, 'FAM2.')
    res2 = blocks(z, P, 'Encoder.1.', num_res, new_buffers, training, tlsc)
    z = basic_conv(res2, P, 'feat_extract.2.', 3, stride=2)
    z = fam(z, z4, P, 'FAM1.')
    z = blocks(z, P, 'Encoder.2.', num_res, new_buffers, training, tlsc)
    z = blocks(z, P, 'Decoder.0.', num_res, new_buffers, training, tlsc)
    o4 = basic_conv(z, P, 'ConvsOut.0.', 3, act=False) + x_4
    z = basic_conv(z, P, 'feat_extract.3.', 4, stride=2, transpose=True)
    z = basic_conv(torch.cat([z, res2], 1), P, 'Convs.0.', 1)
    z = blocks(z, P, 'Decoder.1.', num_res, new_buffers, training, tlsc)
    o2 = basic_conv(z, P, 'ConvsOut.1.', 3, act=False) + x_2
    z = basic_conv(z, P, 'feat_extract.4.', 4, stride=2, transpose=True)
    z = basic_conv(torch.cat([z, res1], 1), P, 'Convs.1.', 1)
    z = blocks(z, P, 'Decoder.2.', num_res, new_buffers, training, tlsc)
    o1 = basic_conv(z, P, 'feat_extract.5.', 3, act=False) + x
    return [o4, o2, o1]

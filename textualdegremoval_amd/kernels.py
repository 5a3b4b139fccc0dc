"""Tensor-level wrappers over the C ABI (include/tdr.h).  PyTorch is plumbing
here: it owns device memory and the HIP stream; all arithmetic happens in
libtdr_hip.so.  Every wrapper enqueues on torch's current stream."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import TdrConvDesc, TdrConvP16Desc, TdrWgradDesc, TdrWgradP16Desc, check

EPI_STD, EPI_GATEBWD, EPI_PSHUF = 0, 1, 2
PACK_FWD, PACK_DGRAD_S1, PACK_DGRAD_2X2S2, PACK_DGRAD_3X3S2 = 0, 1, 2, 3

# Matrix-core arithmetic of the dense convolutions (include/tdr.h, TdrConvDesc.wp_fmt):
#   'bx3' : every fp32 operand split into 3 bf16 terms, 6 cross products on the bf16 MFMA pipe, fp32
#           accumulate -- fp32-equivalent products at 2.7x the fp32-MFMA rate (csrc/tdr_conv_bx3.hip)
#   'f32' : exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain)
#   'hx2' : forward convolutions (PACK_FWD weights: every conv of the forward pass, the frozen ViTs) with a 2-way fp16
#           split, 3 f16 MFMA products per fp32 product -- half the matrix work of 'bx3' at the same measured accuracy
#           (profiles/r1/fp16x2_probe_mi355x.log) for operands inside the fp16 range, which forward activations and weights
#           are; data-gradient and weight-gradient kernels see 1e-7-sized operands and stay on the 3-way bf16 split.
#   'h1'  : plain fp16 MFMA with fp32 accumulation (operands rounded to ONE fp16 plane, one product): the "fp16 MFMA"
#           arithmetic BASELINE configs[4] is quoted on.  Reduced precision (2^-11 per operand), judged on PSNR, never the
#           default; same kernels, weight packs, loss scale and range survey as 'hx2' (the m plane is simply not used).
# The MASA arg-max searches always run on the exact path (near-tie indices must not move).
# Default since round 5: 'bx3' -- the reference's arithmetic (fp32 autograd, image_restoration_ref_model.py:268-279): 24-bit operand
# significands, fp32 exponent range, unscaled gradients, no step guard.  'hx2' is the opt-in FAST MODE (22-bit operands inside the
# fp16 window, loss-scaled backward with a device-resident guard): TDR_MATH=hx2.
MATH = os.environ.get('TDR_MATH', 'bx3')
# weight gradients of 1x1 convs on the split-bf16 kernel as well (0: exact fp32 kernel)
WGRAD_1X1_BX3 = True          # (module switch for A/B runs; not an environment knob)
FMT_F32, FMT_BX3, FMT_HX2, FMT_H1 = 0, 1, 2, 3


def fp16_path():
    """True when the dense contractions take fp16 operands (2-way split or plain): loss scale + range survey apply."""
    return MATH in ('hx2', 'h1')


# True while a backward pass runs on loss-scaled gradients (models/image_restoration_ref_model.py picks the power of two
# that brings dpred to ~2^9): the data-gradient convolutions of TDR_MATH=hx2 then use the fp16 split as well -- every
# gradient operand of a cfg2 step has max|g| within 2^-0.4 .. 2^10 of that scale (profiles/grad_range_survey.py).
GRAD_SCALED = False
# True while the backward pass of a train step runs (tags the operands of kernels.RangeSurvey)
BACKWARD_PHASE = False


def set_grad_scaled(on):
    global GRAD_SCALED
    prev, GRAD_SCALED = GRAD_SCALED, bool(on)
    return prev


def set_math(mode):
    global MATH
    assert mode in ('bx3', 'f32', 'hx2', 'h1')
    MATH = mode


class RangeSurvey:
    """fp16-window survey of one (eager) train step.  While installed (`with survey:`), every operand that is about to
    be split into two fp16 planes -- conv_forward / conv_wgrad inputs on the hx2 path, the inputs of the fused NAFBlock
    kernels -- gets its max |x| recorded on the device (tdr_absmax_bits: one small launch per operand, order-independent
    atomic max).  `read()` brings the table to the host once, after the step.  Operands are tagged 'fwd' (activations,
    weights never leave the window) or 'grad' (operands of the loss-scaled backward pass); the model decides from the two
    exponent ranges whether to move the loss scale or to leave the fp16 split (models/image_restoration_ref_model.py)."""
    SLOTS = 8192

    def __init__(self, device):
        self.table = torch.zeros(self.SLOTS, dtype=torch.int32, device=device)
        self.kinds = []

    def __enter__(self):
        global _survey
        self.prev, _survey = _survey, self
        return self

    def __exit__(self, *exc):
        global _survey
        _survey = self.prev
        return False

    def probe(self, x, kind):
        i = len(self.kinds)
        if i >= self.SLOTS:
            return
        self.kinds.append(kind)
        n = x.shape[0]
        per = x[0].numel()
        ns = x.stride(0) if n > 1 else per
        assert x[0].is_contiguous()
        check(_lib.load().tdr_absmax_bits(x.data_ptr(), ns, n, per, self.table.data_ptr() + 4 * i, _stream()), 'tdr_absmax_bits')

    def read(self):
        """-> {'fwd': (emin, emax, nonfinite), 'grad': (...)}: exponents floor(log2 max|x|) over the non-zero operands"""
        bits = self.table[:len(self.kinds)].cpu().numpy().astype('int64') & 0xffffffff
        out = {}
        for kind in ('fwd', 'grad'):
            b = [int(v) for v, k in zip(bits, self.kinds) if k == kind and v != 0]
            bad = sum(1 for v in b if v >= 0x7f800000)
            e = [((v >> 23) & 0xff) - 127 for v in b if v < 0x7f800000]
            out[kind] = (min(e), max(e), bad) if e else (None, None, bad)
        return out


_survey = None


class PackedWeights:
    """device buffer of packed weights + its format tag"""
    __slots__ = ('buf', 'fmt')

    def __init__(self, buf, fmt):
        self.buf, self.fmt = buf, fmt

    def data_ptr(self):
        return self.buf.data_ptr()


# ---- side stream for work off the critical chain (weight gradients: nothing downstream in the backward needs them).
# Inside `with on_side(...)` every libtdr call is enqueued on the side stream, which first waits for everything
# issued so far on the current stream; allocations still come from the current stream's pool, and every tensor the
# side kernels touch is kept referenced until side_join() (the current stream waits for the side stream), so memory
# is never recycled under a running side kernel.  Under hipGraph capture this becomes a parallel branch of the graph.
# Opt-in (kernels.SIDE_WGRAD = True): measured +1.2 % (NAFNet-ref cfg2) / +2 % (Restormer-ref cfg3) -- the MFMA kernels hold
# their LDS / wave slots while resident, so two of them barely co-run -- and per-kernel profiles of overlapped launches
# are no longer comparable with the serial ones, so the default keeps the backward on one stream.
ATTN_F32 = False              # exact-fp32 attention products inside an fp16 arithmetic (module switch)
ATTN_BX3 = True               # TDR_MATH=bx3: the frozen ViTs' attention on the 3-way bf16 split (False: exact fp32 MFMA, as in rounds 1 - 5)
DWK_GENERIC = False           # the LDS-tiled generic depthwise kernels instead of the register-window ones (module switch: cross-check tests)
SIDE_WGRAD = False            # (module switch: tests/test_hip_network.py::test_side_stream_weight_gradients_match)
_side_stream = None
_side_active = False
_side_dirty = False
_side_keep = []


class on_side:
    def __init__(self, *keep):
        self.keep = keep
        self.on = False

    def __enter__(self):
        global _side_stream, _side_active, _side_dirty
        if SIDE_WGRAD and not _side_active:
            if _side_stream is None:
                _side_stream = torch.cuda.Stream()
            _side_stream.wait_stream(torch.cuda.current_stream())
            _side_active = _side_dirty = self.on = True
            _side_keep.extend(self.keep)
        return self

    def __exit__(self, *exc):
        global _side_active
        if self.on:
            _side_active = False
        return False


def side_keep(*tensors):
    """keep intermediates of the side branch alive until side_join()"""
    if _side_active:
        _side_keep.extend(tensors)
    return tensors[0] if len(tensors) == 1 else tensors


def side_join():
    """make the current stream wait for the side branch (before anything reads its results)"""
    global _side_dirty
    if _side_dirty:
        torch.cuda.current_stream().wait_stream(_side_stream)
        _side_dirty = False
        _side_keep.clear()


class side_suspended:
    """for code that must run on the current stream and read side-branch results even when called from inside an
    on_side block (the gradient sink's bucket gather): joins the side branch, runs the body on the current stream,
    then lets the enclosing on_side block continue on the side stream (after it has re-synchronised)."""

    def __enter__(self):
        global _side_active
        self.was = _side_active
        _side_active = False
        side_join()
        return self

    def __exit__(self, *exc):
        global _side_active, _side_dirty
        if self.was:
            _side_stream.wait_stream(torch.cuda.current_stream())
            _side_active = _side_dirty = True
        return False


# ---- lanes: independent chains of small launches (the 40 MLPs of the stage-A Mapper) on a few HIP streams.  Inside `with lane(i)`
# torch's CURRENT stream is lane i, so this module's launches go there and the caching allocator ties every tensor allocated in
# the block to that stream (stream-ordered reuse stays correct); scratch buffers are per lane.  lanes_join() makes the caller's
# stream wait for all lanes; results that cross to another stream are handed over with record_stream by the caller.
_lane_streams = []
_lane_dirty = {}
_lane_active = False


class lane:
    def __init__(self, i, sync=False):
        """sync: the lane waits for everything issued so far on the current stream at EVERY entry (not only the first since the last
        lanes_join): for work whose operands were produced after the lane's first use"""
        self.i, self.sync = i, sync

    def __enter__(self):
        global _lane_active
        while len(_lane_streams) <= self.i:
            _lane_streams.append(torch.cuda.Stream())
        st = _lane_streams[self.i]
        if self.i not in _lane_dirty or self.sync:
            st.wait_stream(torch.cuda.current_stream())
            _lane_dirty[self.i] = st
        self.ctx = torch.cuda.stream(st)
        self.ctx.__enter__()
        self.prev, _lane_active = _lane_active, True
        return self

    def __exit__(self, *exc):
        global _lane_active
        _lane_active = self.prev
        return self.ctx.__exit__(*exc)


def lanes_join():
    cur = torch.cuda.current_stream()
    for st in _lane_dirty.values():
        cur.wait_stream(st)
    _lane_dirty.clear()


def _stream():
    if _side_active:
        return _side_stream.cuda_stream
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _dense_nchw(t):
    """tensor must be dense in its last 3 dims; returns the image stride."""
    if t.dim() == 4:
        n, c, h, w = t.shape
        assert t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == h * w, 'need NCHW dense per image'
        return t.stride(0) if n > 1 else c * h * w
    raise ValueError('expected 4-D tensor')


def _vec_ns(v):
    """per-channel vector [C] (shared, stride 0) or [N,C] (per image)."""
    if v is None:
        return 0
    return v.shape[-1] if v.dim() == 2 else 0


_ws_cache = {}
_ws_capture_refs = None


class workspace_capture:
    """While a hipGraph is being captured, every scratch buffer handed out is also recorded in `refs`: the graph's
    owner keeps that list next to the graph, so a later eager call with a bigger shape (validation on a full-size
    image between replays) that makes `workspace()` grow and drop its cached buffer cannot free memory the captured
    kernels still address."""

    def __init__(self, refs):
        self.refs = refs

    def __enter__(self):
        global _ws_capture_refs
        self.prev, _ws_capture_refs = _ws_capture_refs, self.refs
        return self

    def __exit__(self, *exc):
        global _ws_capture_refs
        _ws_capture_refs = self.prev
        return False


def workspace(nfloats, device, tag='main'):
    """grow-only scratch buffer (stream-ordered reuse)."""
    key = (tag + ('@side' if _side_active else '') + (f'@lane{torch.cuda.current_stream().cuda_stream}' if _lane_active else ''),
           device.index if hasattr(device, 'index') else device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None and _side_active:
            _side_keep.append(buf)          # a side-stream kernel may still be using the smaller buffer
        buf = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    if _ws_capture_refs is not None:
        _ws_capture_refs.append(buf)
    return buf


def conv_ck(kh_eff):
    return _lib.load().tdr_conv_ck(kh_eff)


def packed_floats(M, Kch, kh_eff):
    return _lib.load().tdr_packed_weight_floats(M, Kch, kh_eff)


def _pack_dims(w, mode):
    Cout, Cin, KH, _ = w.shape
    if mode == PACK_FWD:
        return Cout, Cin, KH
    if mode == PACK_DGRAD_S1:
        return Cin, Cout, KH
    if mode == PACK_DGRAD_2X2S2:
        return 4 * Cin, Cout, 1
    return 4 * Cin, Cout, 2


def _packed_buffer(w, mode, math):
    lib = _lib.load()
    M, Kch, KHe = _pack_dims(w, mode)
    if math in ('hx2', 'h1') and (mode == PACK_FWD or GRAD_SCALED):
        n = lib.tdr_packed_weight_bytes_hx2(M, Kch, KHe) // 4
        fmt = FMT_HX2 if math == 'hx2' else FMT_H1      # h1 reads the head plane of the same pack
    elif math in ('bx3', 'hx2', 'h1'):
        n = lib.tdr_packed_weight_bytes_bx3(M, Kch, KHe) // 4
        fmt = FMT_BX3
    else:
        n = lib.tdr_packed_weight_floats(M, Kch, KHe)
        fmt = FMT_F32
    return PackedWeights(torch.empty(n, dtype=torch.float32, device=w.device), fmt)


class PackPlan:
    """Per-model cache of packed weights.  The first step records every (weight, mode) the engine asks for and
    packs it on the spot into a persistent buffer; from then on `run()` re-packs ALL of them with one
    multi-tensor launch at the start of a step (weights change once per step, in the optimiser) and
    `pack_weights` returns the cached buffers without launching anything."""

    def __init__(self):
        self.entries = {}          # (ptr, shape, mode, math) -> (w, mode, math, PackedWeights)
        self.grouped = {}          # pack_weights_grouped: stacks of per-word matrices whose packs are views of one buffer
        self.table = None
        self.dirty = True
        self.valid = False

    def lookup(self, w, mode, math):
        key = (w.data_ptr(), tuple(w.shape), mode, math, GRAD_SCALED)
        e = self.entries.get(key)
        if e is None:
            pw = _packed_buffer(w, mode, math)
            self.entries[key] = (w, mode, math, pw)
            self.dirty = True
            _pack_into(w, mode, pw)
            return pw
        if not self.valid:
            _pack_into(w, mode, e[3])
        return e[3]

    def run(self):
        """one launch: re-pack every recorded weight (call after the weights changed, before the forward)."""
        if not self.entries:
            return
        lib = _lib.load()
        if self.dirty or self.table is None:
            jobs = (_lib.TdrPackJob * len(self.entries))()
            blocks = 0
            for j, (w, mode, math, pw) in enumerate(self.entries.values()):
                Cout, Cin, KH, _ = w.shape
                check(lib.tdr_pack_job_init(C.byref(jobs[j]), w.data_ptr(), Cout, Cin, KH, mode, min(pw.fmt, FMT_HX2), pw.data_ptr()),
                      'tdr_pack_job_init')
                jobs[j].first_block = blocks
                blocks += (jobs[j].total + 255) // 256
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).clone()
            dev = next(iter(self.entries.values()))[0].device
            self.table = (raw.to(dev), len(self.entries), blocks)
            self.dirty = False
        tab, n, blocks = self.table
        check(lib.tdr_pack_weights_multi(tab.data_ptr(), n, blocks, _stream()), 'tdr_pack_weights_multi')
        self.valid = True

    def invalidate(self):
        self.valid = False


_active_plan = None


def set_pack_plan(plan):
    """install (or clear, with None) the PackPlan consulted by pack_weights; returns the previous one."""
    global _active_plan
    prev = _active_plan
    _active_plan = plan
    return prev


def _pack_into(w, mode, pw):
    lib = _lib.load()
    Cout, Cin, KH, _ = w.shape
    assert w.is_contiguous()
    if pw.fmt in (FMT_HX2, FMT_H1):
        check(lib.tdr_pack_weights_hx2(w.data_ptr(), Cout, Cin, KH, mode, pw.data_ptr(), _stream()), 'tdr_pack_weights_hx2')
    elif pw.fmt == FMT_BX3:
        check(lib.tdr_pack_weights_bx3(w.data_ptr(), Cout, Cin, KH, mode, pw.data_ptr(), _stream()), 'tdr_pack_weights_bx3')
    else:
        check(lib.tdr_pack_weights(w.data_ptr(), Cout, Cin, KH, mode, pw.data_ptr(), _stream()), 'tdr_pack_weights')


def pack_weights(w, mode, out=None, math=None):
    """w: (Cout,Cin,KH,KH) contiguous.  Returns (wp, Mpad, M, Kch, KH_eff); wp is a PackedWeights."""
    M, Kch, KHe = _pack_dims(w, mode)
    math = math or MATH
    if _active_plan is not None and out is None:
        pw = _active_plan.lookup(w, mode, math)
    else:
        pw = PackedWeights(out, FMT_F32) if (out is not None and math == 'f32') else _packed_buffer(w, mode, math)
        _pack_into(w, mode, pw)
    return pw, (M + 31) // 32 * 32, M, Kch, KHe


def pack_weights_grouped(W, mode):
    """W [G, Cout, Cin]: a contiguous stack of G weight matrices (the per-word Linears of the stage-A Mapper).  Returns
    (PackedWeights over G consecutive packs, Mpad, floats per pack) for a G-way grouped tdr_conv_forward (N = G images,
    wp_ns = floats per pack).  Inside a PackPlan the G packs are ordinary plan entries (views into one buffer): `run()` re-packs
    them with the step's one multi-tensor launch."""
    G, Cout, Cin = W.shape
    assert W.is_contiguous()
    plan = _active_plan
    key = ('grouped', W.data_ptr(), (G, Cout, Cin), mode, MATH, GRAD_SCALED)
    store = plan.grouped if plan is not None else None
    e = store.get(key) if store is not None else None
    if e is None:
        w4 = [W[g].view(Cout, Cin, 1, 1) for g in range(G)]
        proto = _packed_buffer(w4[0], mode, MATH)
        per = proto.buf.numel()
        big = torch.empty(G * per, dtype=torch.float32, device=W.device)
        views = [PackedWeights(big[g * per:(g + 1) * per], proto.fmt) for g in range(G)]
        for g in range(G):
            _pack_into(w4[g], mode, views[g])
            if plan is not None:
                plan.entries[(w4[g].data_ptr(), tuple(w4[g].shape), mode, MATH, GRAD_SCALED)] = (w4[g], mode, MATH, views[g])
        M = _pack_dims(w4[0], mode)[0]
        e = (PackedWeights(big, proto.fmt), (M + 31) // 32 * 32, per, w4, views)
        if plan is not None:
            plan.dirty = True
            store[key] = e
    elif not plan.valid:
        for w, v in zip(e[3], e[4]):
            _pack_into(w, mode, v)
    return e[0], e[1], e[2]


def pack_patches(blk, G, PH, PW, pstep, dil, off):
    """blk [B*G, C, BH, BW] -> per-image packed 3x3 patch filters, M = G*PH*PW."""
    lib = _lib.load()
    BG, Cc, BH, BW = blk.shape
    B = BG // G
    M = G * PH * PW
    per_b = lib.tdr_packed_weight_floats(M, Cc, 3)
    wp = torch.empty(B * per_b, dtype=torch.float32, device=blk.device)
    check(lib.tdr_pack_patches(blk.data_ptr(), B, G, Cc, BH, BW, PH, PW, pstep, dil, off, wp.data_ptr(), _stream()),
          'tdr_pack_patches')
    return wp, (M + 31) // 32 * 32, per_b


def conv_forward(x, wp, Mpad, Cout, KH, stride=1, dil=1, pad=0, OH=None, OW=None, out=None, Cin=None, epi=EPI_STD,
                 gate=False, kscale=None, wp_ns=0, bias=None, scale=None, bias2=None, bias2_mul=1.0, res=None,
                 mask=None, aux=None, relu=False):
    lib = _lib.load()
    N, Cx, H, W = x.shape
    if Cin is None:
        Cin = Cx // 2 if gate else Cx
    if OH is None:
        OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    if out is None:
        if epi == EPI_PSHUF:
            out = torch.empty(N, Cout // 4, 2 * OH, 2 * OW, dtype=torch.float32, device=x.device)
        elif epi == EPI_GATEBWD:
            out = torch.empty(N, 2 * Cout, OH, OW, dtype=torch.float32, device=x.device)
        else:
            out = torch.empty(N, Cout, OH, OW, dtype=torch.float32, device=x.device)
    d = TdrConvDesc()
    d.N, d.Cin, d.H, d.W = N, Cin, H, W
    d.Cout, d.OH, d.OW = Cout, OH, OW
    d.KH, d.stride, d.dil, d.pad = KH, stride, dil, pad
    d.inp, d.in_ns = x.data_ptr(), _dense_nchw(x)
    d.gate = 1 if gate else 0
    d.kscale, d.kscale_ns = _p(kscale), _vec_ns(kscale)
    d.wp, d.wp_ns, d.Mpad = wp.data_ptr(), wp_ns, Mpad
    d.wp_fmt = getattr(wp, 'fmt', FMT_F32)
    d.out, d.out_ns = out.data_ptr(), _dense_nchw(out)
    d.epi = epi
    d.bias, d.bias_ns = _p(bias), _vec_ns(bias)
    d.scale, d.scale_ns = _p(scale), _vec_ns(scale)
    d.bias2, d.bias2_ns, d.bias2_mul = _p(bias2), _vec_ns(bias2), float(bias2_mul)
    d.res, d.res_ns = _p(res), (_dense_nchw(res) if res is not None else 0)
    d.mask, d.mask_ns = _p(mask), (_dense_nchw(mask) if mask is not None else 0)
    d.aux, d.aux_ns = _p(aux), (_dense_nchw(aux) if aux is not None else 0)
    d.relu = int(relu) if not isinstance(relu, bool) else (1 if relu else 0)      # 2 = exact GELU
    if _survey is not None and d.wp_fmt in (FMT_HX2, FMT_H1):
        _survey.probe(x, 'grad' if BACKWARD_PHASE else 'fwd')
    check(lib.tdr_conv_forward(C.byref(d), _stream()), 'tdr_conv_forward')
    return out


class P16:
    """A pre-split activation (include/tdr.h, "plane tensors"): 16-byte slots [N][C/8][plane][H+2][W+2] of 8 x 16-bit elements, zero
    border.  fmt FMT_HX2: two fp16 planes (heads, residuals) -- 4 bytes per element, operands inside the fp16 window (TDR_MATH=hx2);
    fmt FMT_BX3: three bf16 planes (h, m, l) -- 6 bytes per element, h + m + l IS the fp32 value (any exponent; TDR_MATH=bx3).
    `buf` is a flat int32 tensor."""
    __slots__ = ('buf', 'N', 'C', 'H', 'W', 'fmt')

    def __init__(self, buf, N, Cc, H, W, fmt=FMT_HX2):
        self.buf, self.N, self.C, self.H, self.W, self.fmt = buf, N, Cc, H, W, fmt

    @staticmethod
    def empty(N, Cc, H, W, device, fmt=FMT_HX2):
        assert Cc % 16 == 0, f'plane tensors need C % 16 == 0 (got {Cc})'
        assert fmt in (FMT_HX2, FMT_BX3)
        n = _lib.load().tdr_p16_bytes_fmt(N, Cc, H, W, fmt) // 4
        return P16(torch.empty(n, dtype=torch.int32, device=device), N, Cc, H, W, fmt)

    def data_ptr(self):
        return self.buf.data_ptr()

    def to_f32(self):
        out = torch.empty(self.N, self.C, self.H, self.W, dtype=torch.float32, device=self.buf.device)
        check(_lib.load().tdr_p16_to_f32_fmt(self.buf.data_ptr(), self.N, self.C, self.H, self.W, out.data_ptr(), _dense_nchw(out),
                                             self.fmt, _stream()), 'tdr_p16_to_f32')
        return out


def p16_supported(Cc):
    """channel counts the pre-split 3x3 path takes"""
    return Cc % 16 == 0 and Cc >= 16


def plane_fmt():
    """plane format of the step's arithmetic: fp16 pairs inside a loss-scaled hx2 step, bf16 triples under bx3, else None"""
    if MATH == 'hx2' and GRAD_SCALED:
        return FMT_HX2
    if MATH == 'bx3':
        return FMT_BX3
    return None


def p16_from_f32(x, out=None, fmt=FMT_HX2):
    N, Cc, H, W = x.shape
    out = out if out is not None else P16.empty(N, Cc, H, W, x.device, fmt)
    if _survey is not None and out.fmt == FMT_HX2:
        _survey.probe(x, 'grad' if BACKWARD_PHASE else 'fwd')
    check(_lib.load().tdr_p16_from_f32_fmt(x.data_ptr(), _dense_nchw(x), N, Cc, H, W, out.data_ptr(), out.fmt, _stream()), 'tdr_p16_from_f32')
    return out


def conv3x3_p16(x16, wp, Mpad, Cout, bias=None, res=None, mask=None, relu=False, want32=True, want16=False, out32=None):
    """3x3 / stride 1 / pad 1 convolution of a P16 tensor (csrc/tdr_conv_p16.hip); `res` / `mask` may be fp32 NCHW tensors or
    P16 tensors.  Returns (fp32 NCHW result or None, P16 result or None)."""
    lib = _lib.load()
    d = TdrConvP16Desc()
    d.N, d.Cin, d.H, d.W, d.Cout = x16.N, x16.C, x16.H, x16.W, Cout
    d.inp = x16.data_ptr()
    d.wp, d.Mpad, d.wp_fmt = wp.data_ptr(), Mpad, getattr(wp, 'fmt', FMT_F32)
    assert d.wp_fmt == x16.fmt, f'weight pack format {d.wp_fmt} does not match the plane format {x16.fmt} of the input'
    assert all(t.fmt == x16.fmt for t in (res, mask) if isinstance(t, P16))
    d.bias = _p(bias)
    if isinstance(res, P16):
        d.res16 = res.data_ptr()
    elif res is not None:
        d.res32, d.res32_ns = res.data_ptr(), _dense_nchw(res)
    if isinstance(mask, P16):
        d.mask16 = mask.data_ptr()
    elif mask is not None:
        d.mask32, d.mask32_ns = mask.data_ptr(), _dense_nchw(mask)
    d.relu = 1 if relu else 0
    o32 = o16 = None
    if want32:
        o32 = out32 if out32 is not None else torch.empty(x16.N, Cout, x16.H, x16.W, dtype=torch.float32, device=x16.buf.device)
        d.out32, d.out32_ns = o32.data_ptr(), _dense_nchw(o32)
    if want16:
        o16 = P16.empty(x16.N, Cout, x16.H, x16.W, x16.buf.device, x16.fmt)
        d.out16 = o16.data_ptr()
    check(lib.tdr_conv3x3_p16(C.byref(d), _stream()), 'tdr_conv3x3_p16')
    if _survey is not None and o16 is not None and o16.fmt == FMT_HX2:
        # the pair planes are the operands of the next contraction: their fp32 value must sit inside the fp16 window
        _survey.probe(o32 if o32 is not None else o16.to_f32(), 'grad' if BACKWARD_PHASE else 'fwd')
    return o32, o16


def wgrad3x3_p16(x16, d16, want_db=False):
    """weight (and bias) gradient of a 3x3 / stride 1 / pad 1 convolution from its P16 input and P16 output gradient
    (csrc/tdr_wgrad_p16.hip).  Returns g [1, Cout, Cin, 3, 3] (and db [Cout])."""
    lib = _lib.load()
    d = TdrWgradP16Desc()
    d.N, d.Cin, d.H, d.W, d.Cout = x16.N, x16.C, x16.H, x16.W, d16.C
    d.in16, d.dout16 = x16.data_ptr(), d16.data_ptr()
    assert x16.fmt == d16.fmt
    d.fmt = x16.fmt
    dev = x16.buf.device
    g = torch.empty(1, d16.C, x16.C, 3, 3, dtype=torch.float32, device=dev)
    db = torch.empty(d16.C, dtype=torch.float32, device=dev) if want_db else None
    d.g, d.db = g.data_ptr(), _p(db)
    ws = workspace(lib.tdr_wgrad3x3_p16_ws_floats(C.byref(d)), dev, 'wgrad')
    d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    check(lib.tdr_wgrad3x3_p16(C.byref(d), _stream()), 'tdr_wgrad3x3_p16')
    return (g, db) if want_db else g


def conv_wgrad(x, dout, Cout, Cin, KH, stride=1, pad=0, gate=False, per_image=False, want_db=False, fp16_range=False):
    """returns G [groups, Cout, Cin, KH, KH] (groups = N if per_image else 1); with want_db also the
    bias gradient sum_{n,y,x} dout [Cout] computed in the same pass."""
    lib = _lib.load()
    N, _, H, W = x.shape
    _, _, OH, OW = dout.shape
    d = TdrWgradDesc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.OH, d.OW = N, Cin, H, W, Cout, OH, OW
    d.KH, d.stride, d.pad = KH, stride, pad
    d.inp, d.in_ns, d.gate = x.data_ptr(), _dense_nchw(x), 1 if gate else 0
    d.dout, d.dout_ns = dout.data_ptr(), _dense_nchw(dout)
    groups = N if per_image else 1
    g = torch.empty(groups, Cout, Cin, KH, KH, dtype=torch.float32, device=x.device)
    d.g = g.data_ptr()
    db = torch.empty(Cout, dtype=torch.float32, device=x.device) if want_db else None
    d.db = _p(db)
    d.per_image = 1 if per_image else 0
    # fp16_range: both operands are forward activations (Restormer's q k^T Gram); otherwise `dout` is a gradient and the
    # fp16 split needs the loss-scaled backward pass (GRAD_SCALED)
    if fp16_path() and (fp16_range or GRAD_SCALED) and (KH == 3 or WGRAD_1X1_BX3):
        d.math = 2 if MATH == 'hx2' else 3
    else:
        d.math = 1 if (MATH != 'f32' and (KH == 3 or WGRAD_1X1_BX3)) else 0
    if _survey is not None and d.math >= 2:
        _survey.probe(dout, 'fwd' if fp16_range else 'grad')
    need = lib.tdr_wgrad_ws_floats(C.byref(d))
    ws = workspace(need, x.device, 'wgrad')
    d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    check(lib.tdr_conv_wgrad(C.byref(d), _stream()), 'tdr_conv_wgrad')
    return (g, db) if want_db else g


def _wgrad1x1_desc(x, dout, Cout, Cin, gate):
    N, _, H, W = x.shape
    d = TdrWgradDesc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.OH, d.OW = N, Cin, H, W, Cout, H, W
    d.KH, d.stride, d.pad = 1, 1, 0
    d.in_ns, d.gate, d.dout_ns = _dense_nchw(x), 1 if gate else 0, _dense_nchw(dout)
    d.per_image = 0
    if fp16_path() and GRAD_SCALED and WGRAD_1X1_BX3:
        d.math = 2 if MATH == 'hx2' else 3
    else:
        d.math = 1 if (MATH != 'f32' and WGRAD_1X1_BX3) else 0
    return d


def wgrad1x1_group_key(x, dout, Cout, Cin, gate):
    """hashable shape signature under which 1x1 weight-gradient requests can share one grouped launch, or None if this one cannot
    (csrc/tdr_wgrad_1x1.hip, tdr_wgrad1x1_group: same N, channels, image size, strides, gate; 16-byte aligned dense operands)"""
    if x.dim() != 4 or dout.dim() != 4 or x.shape[2:] != dout.shape[2:] or x.data_ptr() % 16 or dout.data_ptr() % 16:
        return None
    d = _wgrad1x1_desc(x, dout, Cout, Cin, gate)
    if not _lib.load().tdr_wgrad1x1_group_supported(C.byref(d)):
        return None
    return (d.N, Cin, Cout, d.H, d.W, int(d.in_ns), int(d.dout_ns), int(d.gate), int(d.math))


# Pinned pointer tables of the grouped launches, per call site (`seq` = index of the group within a backward pass) and size.  Allocated
# in EAGER steps only (never inside a stream capture; every shape runs eagerly before it is captured).  The upload node of a hipGraph
# re-reads its pinned block at every replay, so a block handed to a capture is NEVER written again and never handed out twice: every
# eager call of a call site tops its spare list up (to GRP_CAP_SPARE unused blocks), a capture consumes spares and records them with
# the graph's other scratch references (workspace_capture), and a capture that finds no spare raises instead of aliasing a block a
# live graph still reads.  Eager steps alternate between two blocks of their own.
GRP_CAP_SPARE = 4
_grp_pool = {}
_grp_owner = 0


def set_group_owner(token):
    """the pinned-table pools of the grouped launches are per owner (a model passes id(self) at the start of its step): two models of
    one process that replay their own captured graphs never share a pool"""
    global _grp_owner
    _grp_owner = token


def _grp_table(seq, nrows, capturing):
    key = (_grp_owner, seq, nrows)
    ent = _grp_pool.get(key)
    if ent is None:
        if capturing:
            raise RuntimeError('kernels.wgrad1x1_group: this backward pass is being captured before it ever ran eagerly (no pinned table for '
                               f'group {seq}); run one eager step of the shape first or set engine.GROUP_LEAVES = False')
        ent = _grp_pool[key] = dict(eager=[torch.empty(nrows, dtype=torch.int64).pin_memory() for _ in range(2)], ev=[None, None], flip=0,
                                    spare=[])
    if capturing:
        if not ent['spare']:
            raise RuntimeError(f'kernels.wgrad1x1_group: no unused pinned table left for group {seq} ({nrows} rows) -- the captured pass uses '
                               'this call site more often than the eager warm-up steps of the shape did; set engine.GROUP_LEAVES = False')
        host = ent['spare'].pop()
        if _ws_capture_refs is not None:
            _ws_capture_refs.append(host)      # owned by the graph from here on (freed with it)
        return host, None
    if len(ent['spare']) < GRP_CAP_SPARE:
        ent['spare'].append(torch.empty(nrows, dtype=torch.int64).pin_memory())
    slot = ent['flip']
    ent['flip'] ^= 1
    if ent['ev'][slot] is not None:
        ent['ev'][slot].synchronize()          # the upload that last read this block (two eager steps ago) has run
    return ent['eager'][slot], (ent, slot)


def _upload_table(rows, seq, dev):
    """the pointer table of a table-driven launch (call site `seq` of the pass) -> device tensor of 64-bit words, through a pinned host
    block that a captured graph can re-read on replay (_grp_table)"""
    capturing = torch.cuda.is_current_stream_capturing()
    host, eager_slot = _grp_table(seq, len(rows), capturing)
    host.copy_(torch.tensor(rows, dtype=torch.int64))
    tab = torch.empty(len(rows), dtype=torch.int64, device=dev)
    tab.copy_(host, non_blocking=True)
    if eager_slot is not None:
        ent, slot = eager_slot
        ent['ev'][slot] = torch.cuda.Event()
        ent['ev'][slot].record()
    return tab


# ---- table-driven forms of the small finishing reductions of the deferred leaves (one launch per kind instead of one per block;
# ---- per problem bit-identical to the single-problem entry points)
def pair_sum_partials_multi(items, seq=0):
    """items: [(ws, nparts, C)], nparts <= 1024 each (any mix of shapes) -- the per-workgroup LayerNorm-gradient partials of fused NAFBlock
    backward launches (naf_tail_bwd / naf_head_bwd with defer_finish) -> [(gw, gb)]"""
    dev = items[0][0].device
    out = torch.empty(sum(2 * cc for _, _, cc in items), dtype=torch.float32, device=dev)
    rows, res, off = [], [], 0
    for ws, npt, cc in items:
        assert 0 < npt <= 1024
        rows += [ws.data_ptr(), out.data_ptr() + 4 * off, out.data_ptr() + 4 * (off + cc), npt, cc]
        res.append((out[off:off + cc], out[off + cc:off + 2 * cc]))
        off += 2 * cc
    tab = _upload_table(rows, seq, dev)
    check(_lib.load().tdr_pair_sum_partials_multi(tab.data_ptr(), len(items), max(cc for _, _, cc in items), _stream()), 'tdr_pair_sum_partials_multi')
    return res


def dw_param_finish_multi(items, seq=0):
    """items: [(ws, N, C, H, W)] (any mix of shapes; dwsg_bwd with defer_finish) -> [(dw [2C, 1, 3, 3], db [2C])]"""
    lib = _lib.load()
    dev = items[0][0].device
    out = torch.empty(sum(20 * cc for _, _, cc, _, _ in items), dtype=torch.float32, device=dev)
    rows, res, off = [], [], 0
    for ws, N, cc, H, W in items:
        rows += [ws.data_ptr(), out.data_ptr() + 4 * off, out.data_ptr() + 4 * (off + 18 * cc), N, cc, int(lib.tdr_dw_param_finish_nb(H, W))]
        res.append((out[off:off + 18 * cc].view(2 * cc, 1, 3, 3), out[off + 18 * cc:off + 20 * cc]))
        off += 20 * cc
    tab = _upload_table(rows, seq, dev)
    check(lib.tdr_dw_param_finish_multi(tab.data_ptr(), len(items), max(it[2] for it in items), _stream()), 'tdr_dw_param_finish_multi')
    return res


def scaled_conv_param_grads_multi(items, seq=0):
    """items: [(G [Cout, Cin], S [Cout], w, b, gamma)] (any mix of shapes) -> [(dw, db, dgamma)] (scaled_conv_param_grads per item)"""
    dev = items[0][0].device
    shapes = [(it[0].shape[-2], it[0].shape[-1]) for it in items]
    out = torch.empty(sum(co * ci + 2 * co for co, ci in shapes), dtype=torch.float32, device=dev)
    rows, res, off = [], [], 0
    for (G, S, w, b, gamma), (co, ci) in zip(items, shapes):
        assert G.is_contiguous()
        base = out.data_ptr() + 4 * off
        rows += [G.data_ptr(), S.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), base, base + 4 * co * ci, base + 4 * (co * ci + co), co, ci]
        res.append((out[off:off + co * ci].view(co, ci), out[off + co * ci:off + co * ci + co], out[off + co * ci + co:off + co * ci + 2 * co]))
        off += co * ci + 2 * co
    tab = _upload_table(rows, seq, dev)
    check(_lib.load().tdr_scaled_conv_param_grads_multi(tab.data_ptr(), len(items), max(co for co, _ in shapes), _stream()),
          'tdr_scaled_conv_param_grads_multi')
    return res


def wgrad1x1_group(reqs, seq=0, want_db=True):
    """reqs: [(x, dout, Cout, Cin, gate)] of ONE wgrad1x1_group_key.  One launch + one fixed-order reduction for all of them.
    Returns [(g [1, Cout, Cin, 1, 1], db [Cout] or None)] in request order."""
    lib = _lib.load()
    x0, d0, Cout, Cin, gate = reqs[0]
    dev = x0.device
    n = len(reqs)
    d = _wgrad1x1_desc(x0, d0, Cout, Cin, gate)
    per = int(lib.tdr_wgrad1x1_group_ws_floats(C.byref(d), n))
    bpp = per // (Cout * (Cin + 1))
    ws = workspace(per * n, dev, 'wgrad_group')
    g = torch.empty(n, Cout, Cin, dtype=torch.float32, device=dev)
    db = torch.empty(n, Cout, dtype=torch.float32, device=dev) if want_db else None
    rows = []
    for i, (x, do, *_r) in enumerate(reqs):
        part = ws.data_ptr() + 4 * per * i
        rows += [x.data_ptr(), do.data_ptr(), part, part + 4 * bpp * Cout * Cin if want_db else 0, g.data_ptr() + 4 * Cout * Cin * i,
                 db.data_ptr() + 4 * Cout * i if want_db else 0]
    tab = _upload_table(rows, seq, dev)
    if _survey is not None and d.math >= 2:
        for (_x, do, *_r) in reqs:
            _survey.probe(do, 'grad')
    check(lib.tdr_wgrad1x1_group(C.byref(d), n, tab.data_ptr(), _stream()), 'tdr_wgrad1x1_group')
    return [(g[i].view(1, Cout, Cin, 1, 1), db[i] if want_db else None) for i in range(n)]


def layernorm2d_fwd(x, w, b, eps, center=True):
    """center=False (b None): BiasFree_LayerNorm, y = x * rstd * w."""
    lib = _lib.load()
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x.device)
    mu = torch.empty(N, H * W, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mu)
    check(lib.tdr_layernorm2d_fwd(x.data_ptr(), _dense_nchw(x), w.data_ptr(), _p(b), float(eps), 1 if center else 0, N, Cc,
                                  H * W, y.data_ptr(), mu.data_ptr(), rstd.data_ptr(), _stream()), 'tdr_layernorm2d_fwd')
    return y, mu, rstd


def layernorm2d_bwd(go, x, mu, rstd, w, add=None, center=True):
    lib = _lib.load()
    N, Cc, H, W = x.shape
    assert go.is_contiguous()
    gx = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x.device)
    gw = torch.empty(Cc, dtype=torch.float32, device=x.device)
    gb = torch.empty_like(gw)
    ws = workspace(lib.tdr_ln_ws_floats(N, Cc, H * W), x.device)
    add_ns = _dense_nchw(add) if add is not None else 0
    add_C = add.shape[1] if add is not None else 0
    check(lib.tdr_layernorm2d_bwd(go.data_ptr(), x.data_ptr(), _dense_nchw(x), mu.data_ptr(), rstd.data_ptr(),
                                  w.data_ptr(), _p(add), add_ns, add_C, 1 if center else 0, N, Cc, H * W, gx.data_ptr(),
                                  gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), _stream()), 'tdr_layernorm2d_bwd')
    return gx, gw, gb


def dwsg_fwd(t, w, b):
    lib = _lib.load()
    N, C2, H, W = t.shape
    Cc = C2 // 2
    assert t.is_contiguous()
    g = torch.empty(N, Cc, H, W, dtype=torch.float32, device=t.device)
    pooled = torch.empty(N, Cc, dtype=torch.float32, device=t.device)
    ws = workspace(lib.tdr_dwsg_ws_floats(N, Cc, H, W), t.device)
    check(lib.tdr_dwsg_fwd(t.data_ptr(), w.data_ptr(), b.data_ptr(), N, Cc, H, W, g.data_ptr(), pooled.data_ptr(),
                           ws.data_ptr(), _stream()), 'tdr_dwsg_fwd')
    return g, pooled


def dwsg_bwd(dg, t, w, b, dg_bias=None, dg_bias_mul=1.0, defer_finish=False):
    """dg_bias [N, C]: per-plane constant added to dg (times dg_bias_mul) as it is read -- the pooled gradient of the SCA
    branch when the conv3 data gradient comes out of the fused tail kernel without it.
    defer_finish (one-pass backward only; otherwise ignored): returns (dt, fin, None), fin() -> (dw, db) reduces the per-workgroup
    partials of the parameter gradients, which then live in a buffer of their own."""
    lib = _lib.load()
    N, C2, H, W = t.shape
    Cc = C2 // 2
    assert dg.is_contiguous() and t.is_contiguous()
    dt = torch.empty_like(t)
    dev = t.device
    if defer_finish and lib.tdr_dwsg_bwd_parts_supported(W):
        ws = torch.empty(int(lib.tdr_dwsg_ws_floats(N, Cc, H, W)), dtype=torch.float32, device=dev)
        check(lib.tdr_dwsg_bwd_biased(dg.data_ptr(), _p(dg_bias), float(dg_bias_mul), t.data_ptr(), w.data_ptr(), b.data_ptr(), N, Cc,
                                      H, W, dt.data_ptr(), 0, 0, ws.data_ptr(), _stream()), 'tdr_dwsg_bwd')

        def fin():
            dw = torch.empty(C2, 1, 3, 3, dtype=torch.float32, device=dev)
            db = torch.empty(C2, dtype=torch.float32, device=dev)
            check(lib.tdr_dw_param_finish(ws.data_ptr(), N, Cc, H, W, dw.data_ptr(), db.data_ptr(), _stream()), 'tdr_dw_param_finish')
            return dw, db
        fin.batch = ('dw', ws, N, Cc, H, W)
        return dt, fin, None
    dw = torch.empty(C2, 1, 3, 3, dtype=torch.float32, device=dev)
    db = torch.empty(C2, dtype=torch.float32, device=dev)
    ws = workspace(lib.tdr_dwsg_ws_floats(N, Cc, H, W), dev)
    check(lib.tdr_dwsg_bwd_biased(dg.data_ptr(), _p(dg_bias), float(dg_bias_mul), t.data_ptr(), w.data_ptr(), b.data_ptr(), N, Cc,
                                  H, W, dt.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), _stream()), 'tdr_dwsg_bwd')
    return dt, dw, db


def naf_tail_supported(c, hw, c_out=None):
    ok_out = c_out is None or c_out == c or (2 * c_out == c and c_out % 32 == 0)
    # 'hx2': fp16 pair planes in LDS (loss-scaled backward); 'bx3': bf16 triple planes, fp32 range (csrc/tdr_nafblock.hip, SchT)
    return ok_out and MATH in ('hx2', 'bx3') and bool(_lib.load().tdr_naf_tail_supported(int(c), int(hw)))


def naf_tail_fwd(g, s, x, w3p, b3, beta, lnw, lnb, eps, w4p, b4, w5p, b5, gamma, c_out=None):
    """fused conv3 -> +residual -> norm2 -> conv4 -> SimpleGate -> conv5 -> +residual (csrc/tdr_nafblock.hip).
    Returns (out, y, mu2, rs2, yn, t4): exactly the tensors the unfused sequence saves for the backward pass."""
    lib = _lib.load()
    N, Cc, H, W = g.shape
    dev = g.device
    c_out = Cc if c_out is None else c_out
    y = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
    yn = torch.empty_like(y)
    out = torch.empty(N, c_out, H, W, dtype=torch.float32, device=dev)
    t4 = torch.empty(N, 2 * Cc, H, W, dtype=torch.float32, device=dev)
    mu = torch.empty(N, H * W, dtype=torch.float32, device=dev)
    rs = torch.empty_like(mu)
    d = _lib.TdrNafTailDesc()
    d.N, d.C, d.HW, d.eps, d.c_out = N, Cc, H * W, float(eps), c_out
    assert w3p.fmt == w4p.fmt == w5p.fmt
    d.w_fmt = w3p.fmt
    d.g, d.g_ns, d.sca, d.x, d.x_ns = g.data_ptr(), _dense_nchw(g), s.data_ptr(), x.data_ptr(), _dense_nchw(x)
    d.w3, d.w4, d.w5 = w3p.data_ptr(), w4p.data_ptr(), w5p.data_ptr()
    d.b3, d.beta, d.lnw, d.lnb = b3.data_ptr(), beta.data_ptr(), lnw.data_ptr(), lnb.data_ptr()
    d.b4, d.b5, d.gamma = b4.data_ptr(), b5.data_ptr(), gamma.data_ptr()
    d.y, d.y_ns, d.mu, d.rs, d.yn, d.yn_ns = y.data_ptr(), _dense_nchw(y), mu.data_ptr(), rs.data_ptr(), yn.data_ptr(), _dense_nchw(yn)
    d.t4, d.t4_ns, d.out, d.out_ns = t4.data_ptr(), _dense_nchw(t4), out.data_ptr(), _dense_nchw(out)
    if _survey is not None:
        _survey.probe(g, 'fwd')
    check(lib.tdr_naf_tail_fwd(C.byref(d), _stream()), 'tdr_naf_tail_fwd')
    if _survey is not None:                 # the operands formed inside the kernel: yn (LayerNorm output) and the gate product
        _survey.probe(yn, 'fwd')
        _survey.probe(t4, 'fwd')
    return out, y, mu, rs, yn, t4


def _ln_partials_finish(ws, nparts, Cc):
    """closure that reduces the per-workgroup LayerNorm-gradient partials a fused NAFBlock backward left in its PRIVATE `ws`"""
    def fin():
        gw = torch.empty(Cc, dtype=torch.float32, device=ws.device)
        gb = torch.empty_like(gw)
        check(_lib.load().tdr_pair_sum_partials(ws.data_ptr(), nparts, Cc, gw.data_ptr(), gb.data_ptr(),
                                                ws.data_ptr() + 4 * nparts * 2 * Cc, _stream()), 'tdr_pair_sum_partials')
        return gw, gb
    if nparts <= 1024:                       # (the one-stage reduction: what pair_sum_partials_multi batches)
        fin.batch = ('ln', ws, nparts, Cc)
    return fin


def naf_tail_bwd(dout, gamma, t4, y, mu, rs, lnw, w5tp, w4tp, w3tp=None, beta=None, sca=None, defer_finish=False):
    """fused conv5 dgrad -> SimpleGate bwd -> conv4 dgrad -> norm2 bwd (+ skip) (csrc/tdr_nafblock.hip).
    Returns (dy, dt4, gw2, gb2[, dgp]); with w3tp / beta / sca the conv3 data gradient dgp = sca * W3^T (beta * dy) comes out
    of the same launch (without the pooled-gradient term: dwsg_bwd(dg_bias=...) adds it).
    defer_finish: gw2 is a closure -> (gw2, gb2) and gb2 is None -- the reduction of the per-workgroup partials (a leaf) is left to
    the caller, the partials live in a buffer of their own."""
    lib = _lib.load()
    N, Cc, H, W = y.shape
    dev = y.device
    assert dout.is_contiguous()
    dt4 = torch.empty(N, 2 * Cc, H, W, dtype=torch.float32, device=dev)
    dy = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
    nws = lib.tdr_naf_tail_bwd_ws_floats(N, Cc, H * W)
    if defer_finish:
        gw = gb = None
        ws = torch.empty(int(nws), dtype=torch.float32, device=dev)
    else:
        gw = torch.empty(Cc, dtype=torch.float32, device=dev)
        gb = torch.empty_like(gw)
        ws = workspace(nws, dev, 'naftail')
    d = _lib.TdrNafTailBwdDesc()
    d.N, d.C, d.HW, d.c_out = N, Cc, H * W, dout.shape[1]
    assert w5tp.fmt == w4tp.fmt
    d.w_fmt = w5tp.fmt
    d.dout, d.dout_ns, d.gamma = dout.data_ptr(), _dense_nchw(dout), gamma.data_ptr()
    d.t4, d.t4_ns, d.y, d.y_ns = t4.data_ptr(), _dense_nchw(t4), y.data_ptr(), _dense_nchw(y)
    d.mu, d.rs, d.lnw, d.w5t, d.w4t = mu.data_ptr(), rs.data_ptr(), lnw.data_ptr(), w5tp.data_ptr(), w4tp.data_ptr()
    d.dt4, d.dt4_ns, d.dy, d.dy_ns = dt4.data_ptr(), _dense_nchw(dt4), dy.data_ptr(), _dense_nchw(dy)
    d.gw, d.gb, d.ws = _p(gw), _p(gb), ws.data_ptr()
    dgp = None
    if w3tp is not None:
        assert w3tp.fmt == w4tp.fmt and sca.is_contiguous()
        dgp = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
        d.w3t, d.beta, d.sca, d.dgp, d.dgp_ns = w3tp.data_ptr(), beta.data_ptr(), sca.data_ptr(), dgp.data_ptr(), _dense_nchw(dgp)
    if _survey is not None:
        _survey.probe(dout, 'grad')
    check(lib.tdr_naf_tail_bwd(C.byref(d), _stream()), 'tdr_naf_tail_bwd')
    if _survey is not None:
        _survey.probe(dt4, 'grad')            # the K = 2C operand formed inside the kernel
        if dgp is not None:
            _survey.probe(dy, 'grad')         # (times beta: the K = C operand of the conv3 stage)
    if defer_finish:
        gw = _ln_partials_finish(ws, N * (H * W // 64), Cc)
    return (dy, dt4, gw, gb) if dgp is None else (dy, dt4, gw, gb, dgp)


def naf_head_fwd(x, lnw, lnb, eps, w1p, b1):
    """fused norm1 -> conv1.  Returns (xn, mu, rs, t1)."""
    lib = _lib.load()
    N, Cc, H, W = x.shape
    dev = x.device
    xn = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
    t1 = torch.empty(N, 2 * Cc, H, W, dtype=torch.float32, device=dev)
    mu = torch.empty(N, H * W, dtype=torch.float32, device=dev)
    rs = torch.empty_like(mu)
    d = _lib.TdrNafHeadFwdDesc()
    d.N, d.C, d.HW, d.w_fmt = N, Cc, H * W, w1p.fmt
    d.x, d.x_ns, d.lnw, d.lnb, d.eps = x.data_ptr(), _dense_nchw(x), lnw.data_ptr(), lnb.data_ptr(), float(eps)
    d.w1, d.b1 = w1p.data_ptr(), b1.data_ptr()
    d.mu, d.rs, d.xn, d.xn_ns, d.t1, d.t1_ns = mu.data_ptr(), rs.data_ptr(), xn.data_ptr(), _dense_nchw(xn), t1.data_ptr(), _dense_nchw(t1)
    if _survey is not None:
        _survey.probe(x, 'fwd')
    check(lib.tdr_naf_head_fwd(C.byref(d), _stream()), 'tdr_naf_head_fwd')
    return xn, mu, rs, t1


def naf_head_bwd(dt1, x, mu, rs, lnw, w1tp, res, defer_finish=False):
    """fused conv1 dgrad -> norm1 bwd (+ skip gradient `res`).  Returns (dx, gw1, gb1); defer_finish as in naf_tail_bwd."""
    lib = _lib.load()
    N, Cc, H, W = x.shape
    dev = x.device
    assert dt1.is_contiguous()
    dx = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
    nws = lib.tdr_naf_tail_bwd_ws_floats(N, Cc, H * W)
    if defer_finish:
        gw = gb = None
        ws = torch.empty(int(nws), dtype=torch.float32, device=dev)
    else:
        gw = torch.empty(Cc, dtype=torch.float32, device=dev)
        gb = torch.empty_like(gw)
        ws = workspace(nws, dev, 'naftail')
    d = _lib.TdrNafHeadBwdDesc()
    d.N, d.C, d.HW, d.w_fmt = N, Cc, H * W, w1tp.fmt
    d.dt1, d.dt1_ns, d.x, d.x_ns = dt1.data_ptr(), _dense_nchw(dt1), x.data_ptr(), _dense_nchw(x)
    d.mu, d.rs, d.lnw, d.w1t = mu.data_ptr(), rs.data_ptr(), lnw.data_ptr(), w1tp.data_ptr()
    d.res, d.res_ns, d.dx, d.dx_ns = res.data_ptr(), _dense_nchw(res), dx.data_ptr(), _dense_nchw(dx)
    d.gw, d.gb, d.ws = _p(gw), _p(gb), ws.data_ptr()
    if _survey is not None:
        _survey.probe(dt1, 'grad')
    check(lib.tdr_naf_head_bwd(C.byref(d), _stream()), 'tdr_naf_head_bwd')
    if defer_finish:
        gw = _ln_partials_finish(ws, N * (H * W // 64), Cc)
    return dx, gw, gb


def sca_fwd(pooled, wsca, bsca):
    lib = _lib.load()
    N, Cc = pooled.shape
    s = torch.empty(N, Cc, dtype=torch.float32, device=pooled.device)
    check(lib.tdr_sca_fwd(pooled.data_ptr(), wsca.data_ptr(), bsca.data_ptr(), N, Cc, s.data_ptr(), _stream()), 'tdr_sca_fwd')
    return s


def sca_bwd(G3, S3, w3, b3, beta, s, pooled, wsca):
    lib = _lib.load()
    N, Cc = s.shape
    dev = s.device
    dw3 = torch.empty(Cc, Cc, 1, 1, dtype=torch.float32, device=dev)
    db3 = torch.empty(Cc, dtype=torch.float32, device=dev)
    dbeta = torch.empty(1, Cc, 1, 1, dtype=torch.float32, device=dev)
    dwsca = torch.empty(Cc, Cc, 1, 1, dtype=torch.float32, device=dev)
    dbsca = torch.empty(Cc, dtype=torch.float32, device=dev)
    dpooled = torch.empty(N, Cc, dtype=torch.float32, device=dev)
    ws = torch.empty(N * Cc, dtype=torch.float32, device=dev)
    check(lib.tdr_sca_bwd(G3.data_ptr(), S3.data_ptr(), w3.data_ptr(), b3.data_ptr(), beta.data_ptr(), s.data_ptr(),
                          pooled.data_ptr(), wsca.data_ptr(), N, Cc, dw3.data_ptr(), db3.data_ptr(), dbeta.data_ptr(),
                          dwsca.data_ptr(), dbsca.data_ptr(), dpooled.data_ptr(), ws.data_ptr(), _stream()), 'tdr_sca_bwd')
    return dw3, db3, dbeta, dwsca, dbsca, dpooled


def scaled_conv_param_grads(G, S, w, b, gamma):
    """G [Cout,Cin], S [Cout]; w/b/gamma may have more rows than Cout (first rows are used)."""
    lib = _lib.load()
    Cout, Cin = G.shape[-2], G.shape[-1]
    dev = G.device
    dw = torch.empty(Cout, Cin, dtype=torch.float32, device=dev)
    db = torch.empty(Cout, dtype=torch.float32, device=dev)
    dg = torch.empty(Cout, dtype=torch.float32, device=dev)
    check(lib.tdr_scaled_conv_param_grads(G.data_ptr(), S.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), Cout,
                                          Cin, dw.data_ptr(), db.data_ptr(), dg.data_ptr(), _stream()),
          'tdr_scaled_conv_param_grads')
    return dw, db, dg


def channel_sum(x):
    lib = _lib.load()
    N, Cc, H, W = x.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = workspace(lib.tdr_chansum_ws_floats(N, Cc, H * W), x.device)
    check(lib.tdr_channel_sum(x.data_ptr(), _dense_nchw(x), N, Cc, H * W, out.data_ptr(), ws.data_ptr(), _stream()),
          'tdr_channel_sum')
    return out


def multi_copy(src_tab, dst_tab, sizes, chunk_tensor, chunk_index, n_chunks, scale=1.0, guard=None):
    """device pointer tables (int64 tensors); one launch copies (and scales) every listed tensor.  With a StepGuard the
    scale is its device-resident 1 / loss scale."""
    if guard is not None:
        check(_lib.load().tdr_multi_copy_guarded(src_tab.data_ptr(), dst_tab.data_ptr(), sizes.data_ptr(), chunk_tensor.data_ptr(),
                                                 chunk_index.data_ptr(), n_chunks, guard.data_ptr(), _stream()),
              'tdr_multi_copy_guarded')
        return
    check(_lib.load().tdr_multi_copy(src_tab.data_ptr(), dst_tab.data_ptr(), sizes.data_ptr(), chunk_tensor.data_ptr(),
                                     chunk_index.data_ptr(), n_chunks, float(scale), _stream()), 'tdr_multi_copy')


def copy_rows(src, src_ns, dst, dst_ns, N, length):
    check(_lib.load().tdr_copy_rows(src.data_ptr(), src_ns, dst.data_ptr(), dst_ns, N, length, _stream()), 'tdr_copy_rows')


def add_rows(src, src_ns, dst, dst_ns, N, length):
    check(_lib.load().tdr_add_rows(src.data_ptr(), src_ns, dst.data_ptr(), dst_ns, N, length, _stream()), 'tdr_add_rows')


def concat2(a, b):
    """cat([a,b], dim=1) with the library's copy kernel (:719,727)."""
    N, Ca, H, W = a.shape
    Cb = b.shape[1]
    out = torch.empty(N, Ca + Cb, H, W, dtype=torch.float32, device=a.device)
    tot = (Ca + Cb) * H * W
    copy_rows(a, _dense_nchw(a), out, tot, N, Ca * H * W)
    copy_rows(b, _dense_nchw(b), out[:, Ca:], tot, N, Cb * H * W)
    return out


def slice_channels(x, c0, c1):
    """contiguous copy of x[:, c0:c1]."""
    N, Cc, H, W = x.shape
    out = torch.empty(N, c1 - c0, H, W, dtype=torch.float32, device=x.device)
    copy_rows(x[:, c0:c1], _dense_nchw(x), out, (c1 - c0) * H * W, N, (c1 - c0) * H * W)
    return out


def add_(dst, src):
    """dst += src (same shape, dense per image)."""
    N = dst.shape[0]
    per = dst[0].numel()
    add_rows(src, _dense_nchw(src) if src.dim() == 4 else per, dst, _dense_nchw(dst) if dst.dim() == 4 else per, N, per)
    return dst


def pixel_unshuffle2(x):
    N, Cc, H2, W2 = x.shape
    assert x.is_contiguous()
    out = torch.empty(N, 4 * Cc, H2 // 2, W2 // 2, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_pixel_unshuffle2(x.data_ptr(), N, Cc, H2 // 2, W2 // 2, out.data_ptr(), _stream()),
          'tdr_pixel_unshuffle2')
    return out


def pad_crop(x, Hd, Wd):
    N, Cc, Hs, Ws = x.shape
    assert x.is_contiguous()
    out = torch.empty(N, Cc, Hd, Wd, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_pad_crop(x.data_ptr(), N, Cc, Hs, Ws, out.data_ptr(), Hd, Wd, _stream()), 'tdr_pad_crop')
    return out


def relu_bwd(go, act):
    assert go.is_contiguous() and act.is_contiguous()
    out = torch.empty_like(go)
    check(_lib.load().tdr_relu_bwd(go.data_ptr(), act.data_ptr(), go.numel(), out.data_ptr(), _stream()), 'tdr_relu_bwd')
    return out


def l1_loss(pred, target, loss_weight=1.0, grad_scale=1.0, guard=None):
    """guard: a StepGuard whose device-resident loss scale multiplies dpred (instead of the host value grad_scale)"""
    assert pred.is_contiguous() and target.is_contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred)
    ws = workspace(4096, pred.device, 'l1')
    if guard is not None:
        check(_lib.load().tdr_l1_loss_guarded(pred.data_ptr(), target.data_ptr(), pred.numel(), float(loss_weight), guard.data_ptr(),
                                              loss.data_ptr(), dpred.data_ptr(), ws.data_ptr(), _stream()), 'tdr_l1_loss_guarded')
        return loss, dpred
    check(_lib.load().tdr_l1_loss(pred.data_ptr(), target.data_ptr(), pred.numel(), float(loss_weight), float(grad_scale), loss.data_ptr(),
                                  dpred.data_ptr(), ws.data_ptr(), _stream()), 'tdr_l1_loss')
    return loss, dpred


LOSS_L1, LOSS_MSE, LOSS_CHARBONNIER, LOSS_PSNR, LOSS_PSNR_Y = range(5)


def pixel_loss(kind, pred, target, loss_weight=1.0, eps=1e-3, grad_scale=1.0, guard=None):
    """pred / target [N,C,H,W] -> (loss [1], dpred): the criteria of losses/losses.py (LOSS_* kinds), mean reduction"""
    assert pred.is_contiguous() and target.is_contiguous() and pred.shape == target.shape and pred.dim() == 4
    N, Cc, H, W = pred.shape
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred)
    ws = workspace(4096 + 128 * N, pred.device, 'l1')
    check(_lib.load().tdr_pixel_loss(int(kind), pred.data_ptr(), target.data_ptr(), N, Cc * H * W, H * W, float(loss_weight), float(eps),
                                     float(grad_scale), guard.data_ptr() if guard is not None else None, loss.data_ptr(),
                                     dpred.data_ptr(), ws.data_ptr(), _stream()), 'tdr_pixel_loss')
    return loss, dpred


# ------------------------------------------------------------------ MASA
def lr_blocks_fwd(feat, py, px, ky, kx):
    N, Cc, H, W = feat.shape
    assert feat.is_contiguous()
    blk = torch.empty(N * py * px, Cc, ky + 2, kx + 2, dtype=torch.float32, device=feat.device)
    check(_lib.load().tdr_lr_blocks_fwd(feat.data_ptr(), N, Cc, H, W, py, px, ky, kx, blk.data_ptr(), _stream()),
          'tdr_lr_blocks_fwd')
    return blk


def lr_blocks_bwd(dblk, N, Cc, H, W, py, px, ky, kx):
    dfeat = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dblk.device)
    check(_lib.load().tdr_lr_blocks_bwd(dblk.data_ptr(), N, Cc, H, W, py, px, ky, kx, dfeat.data_ptr(), _stream()),
          'tdr_lr_blocks_bwd')
    return dfeat


def patch_inv_norm(x, OH, OW, dil=1, pad=0, step=1, off=0, out=None):
    B, Cc, H, W = x.shape
    assert x.is_contiguous()
    inv = out if out is not None else torch.empty(B, OH, OW, dtype=torch.float32, device=x.device)
    assert inv.is_contiguous() and inv.numel() == B * OH * OW
    check(_lib.load().tdr_patch_inv_norm(x.data_ptr(), B, Cc, H, W, OH, OW, dil, pad, step, off, inv.data_ptr(), _stream()),
          'tdr_patch_inv_norm')
    return inv


def coarse_argmax_box(dots, invq, invk, N, P, Hr, Wr, diameter):
    ND = dots.shape[0]
    dev = dots.device
    index = torch.empty(N * P, dtype=torch.int32, device=dev)
    y1 = torch.empty_like(index)
    x1 = torch.empty_like(index)
    check(_lib.load().tdr_coarse_argmax_box(dots.data_ptr(), invq.data_ptr(), invk.data_ptr(), ND, N, P, Hr, Wr, diameter,
                                            index.data_ptr(), y1.data_ptr(), x1.data_ptr(), _stream()),
          'tdr_coarse_argmax_box')
    return index, y1, x1


def gather_ref_block(feat, y1, x1, P, side, s):
    N, Cc, H, W = feat.shape
    out = torch.empty(N * P, Cc, side * s, side * s, dtype=torch.float32, device=feat.device)
    check(_lib.load().tdr_gather_ref_block(feat.data_ptr(), N, Cc, H, W, y1.data_ptr(), x1.data_ptr(), P, side, s,
                                           out.data_ptr(), _stream()), 'tdr_gather_ref_block')
    return out


def scatter_ref_block(dblk, dfeat, y1, x1, P, side):
    N, Cc, H, W = dfeat.shape
    check(_lib.load().tdr_scatter_ref_block(dblk.data_ptr(), N, Cc, H, W, y1.data_ptr(), x1.data_ptr(), P, side,
                                            dfeat.data_ptr(), _stream()), 'tdr_scatter_ref_block')


def fine_argmax(dots, invq, invk, B, P, R):
    dev = dots.device
    index_all = torch.empty(B, P, dtype=torch.int32, device=dev)
    soft_att = torch.empty(B, P, dtype=torch.float32, device=dev)
    check(_lib.load().tdr_fine_argmax(dots.data_ptr(), invq.data_ptr(), invk.data_ptr(), B, P, R, index_all.data_ptr(),
                                      soft_att.data_ptr(), _stream()), 'tdr_fine_argmax')
    return index_all, soft_att


def fine_search_bwd(datt, soft_att, index_all, lrb, refb, invq, invk, K, D):
    B, Cc = lrb.shape[0], lrb.shape[1]
    dlrb = torch.empty_like(lrb)
    drefb = torch.empty_like(refb)
    check(_lib.load().tdr_fine_search_bwd(datt.data_ptr(), soft_att.data_ptr(), index_all.data_ptr(), lrb.data_ptr(),
                                          refb.data_ptr(), invq.data_ptr(), invk.data_ptr(), B, Cc, K, D, dlrb.data_ptr(),
                                          drefb.data_ptr(), _stream()), 'tdr_fine_search_bwd')
    return dlrb, drefb


def transfer_fwd(feat, y1, x1, index_all, soft_att, py, px, K, side, s, out=None):
    N, Cc, H, W = feat.shape
    assert feat.is_contiguous()
    if out is None:
        out = torch.empty(N, Cc, py * K * s, px * K * s, dtype=torch.float32, device=feat.device)
    check(_lib.load().tdr_transfer_fwd(feat.data_ptr(), N, Cc, H, W, y1.data_ptr(), x1.data_ptr(), index_all.data_ptr(),
                                       soft_att.data_ptr(), py, px, K, side, s, out.data_ptr(), _dense_nchw(out),
                                       _stream()), 'tdr_transfer_fwd')
    return out


# TDR_DETERMINISTIC=1: the MASA transfer backward accumulates its scatter in 64-bit fixed point (bit-identical results from
# run to run; every other kernel of the step is deterministic already).  Default off: float atomics, +0.5 ms faster per step.
DETERMINISTIC = os.environ.get('TDR_DETERMINISTIC', '0') == '1'


def transfer_bwd(dout, feat, y1, x1, index_all, soft_att, py, px, K, side, s, dfeat, datt):
    N, Cc, H, W = feat.shape
    ws = workspace(_lib.load().tdr_transfer_ws_floats(N, Cc, H, W, py, px, K, s), feat.device, 'transfer')
    check(_lib.load().tdr_transfer_bwd(dout.data_ptr(), _dense_nchw(dout), feat.data_ptr(), N, Cc, H, W, y1.data_ptr(),
                                       x1.data_ptr(), index_all.data_ptr(), soft_att.data_ptr(), py, px, K, side, s,
                                       1 if DETERMINISTIC else 0, dfeat.data_ptr(), datt.data_ptr(), ws.data_ptr(), _stream()),
          'tdr_transfer_bwd')


# ------------------------------------------------------------------ frozen ViT window matcher (DINOv2)
def resize_bilinear(x, Hd, Wd):
    B, Cc, Hs, Ws = x.shape
    assert x.is_contiguous()
    out = torch.empty(B, Cc, Hd, Wd, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_resize_bilinear(x.data_ptr(), B * Cc, Hs, Ws, out.data_ptr(), Hd, Wd, _stream()), 'tdr_resize_bilinear')
    return out


def unfold_windows(ref, h, stride):
    B, Cc, Hr, Wr = ref.shape
    assert ref.is_contiguous()
    N = ((Hr - h) // stride + 1) * ((Wr - h) // stride + 1)
    out = torch.empty(B * N, Cc, h, h, dtype=torch.float32, device=ref.device)
    check(_lib.load().tdr_unfold_windows(ref.data_ptr(), B, Cc, Hr, Wr, h, stride, out.data_ptr(), _stream()), 'tdr_unfold_windows')
    return out, N


def token_ld(T):
    """padded token-row length: class token + T patches, rounded up to a multiple of 32"""
    return (T + 1 + 31) // 32 * 32


def patchify(x, p, flat=False):
    """-> [B, Ci*p*p, LD/32, 32] with patch t at flat column 1+t (zeros at column 0 and in the padding);
    flat: the batch-flattened layout [1, Ci*p*p, B*LD/32, 32] (image b's tokens at columns b*LD ..)"""
    B, Ci, H, W = x.shape
    assert x.is_contiguous()
    T = (H // p) * (W // p)
    LD = token_ld(T)
    shape = (1, Ci * p * p, B * LD // 32, 32) if flat else (B, Ci * p * p, LD // 32, 32)
    out = torch.empty(*shape, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_patchify(x.data_ptr(), B, Ci, H, W, p, LD, 1 if flat else 0, out.data_ptr(), _stream()), 'tdr_patchify')
    return out, T


def vit_assemble_(tok, cls, pos_cm, T, flat_batch=0):
    """flat_batch = B for a batch-flattened token tensor [1, D, B*LD/32, 32], 0 for [B, D, LD/32, 32]"""
    D = tok.shape[1]
    B = flat_batch or tok.shape[0]
    LD = tok.shape[2] * tok.shape[3] // (flat_batch or 1)
    check(_lib.load().tdr_vit_assemble(tok.data_ptr(), cls.data_ptr(), pos_cm.data_ptr(), B, D, T, LD, 1 if flat_batch else 0, _stream()),
          'tdr_vit_assemble')
    return tok


def attention_fwd(qkv, heads, scale, T1, flat_batch=0, single_product=False):
    """qkv [B, 3C, LD/32, 32] (or batch-flattened [1, 3C, B*LD/32, 32] with flat_batch = B); attends over the first T1 columns
    of every image"""
    B, C3 = (flat_batch or qkv.shape[0]), qkv.shape[1]
    Cc, LD = C3 // 3, qkv.shape[2] * qkv.shape[3] // (flat_batch or 1)
    assert qkv.is_contiguous()
    out = torch.empty(qkv.shape[0], Cc, qkv.shape[2], qkv.shape[3], dtype=torch.float32, device=qkv.device)
    # the frozen ViTs (no gradient flows through this attention): on the fp16 split whenever the dense contractions are
    math = 2 if (MATH in ('hx2', 'h1') and not ATTN_F32) else (1 if (MATH == 'bx3' and ATTN_BX3) else 0)
    if single_product and math == 2 and Cc // heads in (16, 32, 64):
        math = 3          # plain fp16 products (the DINOv2 matcher: only an arg-max leaves it)
    check(_lib.load().tdr_attention_fwd_math(qkv.data_ptr(), B, Cc, heads, T1, LD, float(scale), math, 1 if flat_batch else 0, out.data_ptr(),
                                             _stream()),
          'tdr_attention_fwd')
    return out


# ---- token-major fp16 pipeline of the frozen DINOv2 matcher (csrc/tdr_tok16.hip) ------------------------------------------
def transpose_f32(src):
    """[B, R, C] fp32 -> [B, C, R]"""
    B, R, Cc = src.shape
    assert src.is_contiguous() and src.dtype == torch.float32
    dst = torch.empty(B, Cc, R, dtype=torch.float32, device=src.device)
    check(_lib.load().tdr_transpose_f32(src.data_ptr(), B, R, Cc, dst.data_ptr(), _stream()), 'tdr_transpose_f32')
    return dst


def tok_layernorm(x, w, b, eps, out_f16=True, planes=False):
    """nn.LayerNorm over the rows of x [P, D] (fp32) -> fp16 (a GEMM operand), fp32 [P, D], or (planes) the 2-way split [2, P, D] fp16
    (planes=True / 2) or the 3-way split [3, P, D] bf16 (planes=3)"""
    P, D = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    planes = 2 if planes is True else int(planes)
    if planes == 3:
        out = torch.empty(3, P, D, dtype=torch.bfloat16, device=x.device)
    elif planes:
        out = torch.empty(2, P, D, dtype=torch.float16, device=x.device)
    else:
        out = torch.empty(P, D, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    check(_lib.load().tdr_tok_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), P, D, float(eps), planes if planes else (1 if out_f16 else 0),
                                        out.data_ptr(), _stream()), 'tdr_tok_layernorm')
    return out


def split_planes(w):
    """host-side 2-way split of a frozen fp32 matrix [N, K] -> hi | lo fp16 planes [2, N, K]"""
    hi = w.to(torch.float16)
    return torch.stack([hi, (w - hi.to(torch.float32)).to(torch.float16)]).contiguous()


def split_planes3(w):
    """host-side 3-way split of a frozen fp32 matrix [N, K] -> h | m | l bf16 planes [3, N, K] (round-to-nearest conversions, exact
    fp32 subtractions: the bits of tdr_split3_bf16; h + m + l == w)"""
    h = w.to(torch.bfloat16)
    r = w - h.to(torch.float32)
    m = r.to(torch.bfloat16)
    return torch.stack([h, m, (r - m.to(torch.float32)).to(torch.bfloat16)]).contiguous()


def tok16x3_gemm(x3, w3, bias, epi, act=0, out32=None, ls=None):
    """x3 [3, P, K] . w3 [3, N, K]^T on the 3-way bf16 split (+ bias): epi 2 -> out32 [P, N] += ls * ., in place; 3 -> fp32 [N, P]
    (channel-major); 4 -> split(act(.)) [3, P, N]"""
    _, P, Kd = x3.shape
    N = w3.shape[1]
    assert x3.is_contiguous() and w3.is_contiguous() and x3.dtype == torch.bfloat16 and w3.dtype == torch.bfloat16 and w3.shape[2] == Kd
    assert x3.shape[0] == 3 and w3.shape[0] == 3
    y = None
    if epi == 2:
        assert out32 is not None and out32.is_contiguous() and out32.dtype == torch.float32 and tuple(out32.shape) == (P, N)
    elif epi == 3:
        out32 = torch.empty(N, P, dtype=torch.float32, device=x3.device)
    else:
        y = torch.empty(3, P, N, dtype=torch.bfloat16, device=x3.device)
    check(_lib.load().tdr_tok16x3_gemm(x3.data_ptr(), w3.data_ptr(), _p(bias), P, N, Kd, int(epi), int(act), _p(y), _p(out32), _p(ls),
                                       _stream()), 'tdr_tok16x3_gemm')
    return y if epi == 4 else out32


def cm_to_tok16x3(src):
    """fp32 channel-major [C, P] -> token-major h | m | l bf16 planes [3, P, C]"""
    Cc, P = src.shape
    assert src.is_contiguous() and src.dtype == torch.float32
    out = torch.empty(3, P, Cc, dtype=torch.bfloat16, device=src.device)
    check(_lib.load().tdr_cm_to_tok16x3(src.data_ptr(), Cc, P, out.data_ptr(), _stream()), 'tdr_cm_to_tok16x3')
    return out


def tok16x2_gemm(x2, w2, bias, epi, act=0, out32=None):
    """x2 [2, P, K] . w2 [2, N, K]^T on the 2-way split (+ bias): epi 2 -> out32 [P, N] += ., in place; 3 -> fp32 [N, P] (channel-major);
    4 -> split(act(.)) [2, P, N]"""
    _, P, Kd = x2.shape
    N = w2.shape[1]
    assert x2.is_contiguous() and w2.is_contiguous() and x2.dtype == torch.float16 and w2.dtype == torch.float16 and w2.shape[2] == Kd
    y = None
    if epi == 2:
        assert out32 is not None and out32.is_contiguous() and out32.dtype == torch.float32 and tuple(out32.shape) == (P, N)
    elif epi == 3:
        out32 = torch.empty(N, P, dtype=torch.float32, device=x2.device)
    else:
        y = torch.empty(2, P, N, dtype=torch.float16, device=x2.device)
    check(_lib.load().tdr_tok16x2_gemm(x2.data_ptr(), w2.data_ptr(), _p(bias), P, N, Kd, int(epi), int(act), _p(y), _p(out32), _stream()),
          'tdr_tok16x2_gemm')
    return y if epi == 4 else out32


def cm_to_tok16x2(src):
    """fp32 channel-major [C, P] -> token-major split planes [2, P, C]"""
    Cc, P = src.shape
    assert src.is_contiguous() and src.dtype == torch.float32
    out = torch.empty(2, P, Cc, dtype=torch.float16, device=src.device)
    check(_lib.load().tdr_cm_to_tok16x2(src.data_ptr(), Cc, P, out.data_ptr(), _stream()), 'tdr_cm_to_tok16x2')
    return out


def tok16_gemm(x16, w16, bias, epi=0, res=None, ls=None):
    """x16 [P, K] . w16 [N, K]^T (+ bias): epi 0 -> fp16 [P, N]; 1 -> erf-GELU, fp16; 2 -> res [P, N] (fp32) += ls * (. + bias), in place"""
    P, Kd = x16.shape
    N = w16.shape[0]
    assert x16.is_contiguous() and w16.is_contiguous() and x16.dtype == torch.float16 and w16.dtype == torch.float16 and w16.shape[1] == Kd
    y = None
    if epi == 2:
        assert res is not None and res.is_contiguous() and res.dtype == torch.float32 and tuple(res.shape) == (P, N)
    else:
        y = torch.empty(P, N, dtype=torch.float16, device=x16.device)
    check(_lib.load().tdr_tok16_gemm(x16.data_ptr(), w16.data_ptr(), _p(bias), P, N, Kd, int(epi), _p(y), _p(res), _p(ls), _stream()),
          'tdr_tok16_gemm')
    return res if epi == 2 else y


def tok16_attention(qkv16, B, heads, scale, T1):
    """qkv16 [B * LD, 3C] fp16 (token-major) -> [B * LD, C] fp16; attends over the first T1 rows of every image"""
    P, C3 = qkv16.shape
    assert qkv16.is_contiguous() and qkv16.dtype == torch.float16 and P % B == 0
    out = torch.empty(P, C3 // 3, dtype=torch.float16, device=qkv16.device)
    check(_lib.load().tdr_tok16_attention(qkv16.data_ptr(), B, C3 // 3, heads, T1, P // B, float(scale), out.data_ptr(), _stream()),
          'tdr_tok16_attention')
    return out


def token_match(fl, fr, windows, N, T1):
    """fl [B,D,LD/32,32], fr [B*N,D,LD/32,32], windows [B*N,C,h,w] -> (corr [B,N], index [B] int32, ref_in [B,C,h,w])"""
    B, D = fl.shape[0], fl.shape[1]
    LD = fl.shape[2] * fl.shape[3]
    per = windows[0].numel()
    corr = torch.empty(B, N, dtype=torch.float32, device=fl.device)
    index = torch.empty(B, dtype=torch.int32, device=fl.device)
    ref_in = torch.empty((B,) + tuple(windows.shape[1:]), dtype=torch.float32, device=fl.device)
    check(_lib.load().tdr_token_match(fl.data_ptr(), fr.data_ptr(), windows.data_ptr(), B, N, D, T1, LD, per, corr.data_ptr(),
                                      index.data_ptr(), ref_in.data_ptr(), _stream()), 'tdr_token_match')
    return corr, index, ref_in


# ------------------------------------------------------------------ Restormer-ref (network_restormer_guided_arch.py)
def dwgelu_fwd(t, w, b=None):
    """GDFN gate: gelu(dw(t)[:, :h]) * dw(t)[:, h:]"""
    N, C2, H, W = t.shape
    assert t.is_contiguous()
    g = torch.empty(N, C2 // 2, H, W, dtype=torch.float32, device=t.device)
    check(_lib.load().tdr_dwgelu_fwd(t.data_ptr(), w.data_ptr(), _p(b), N, C2 // 2, H, W, g.data_ptr(), _stream()), 'tdr_dwgelu_fwd')
    return g


def dwgelu_bwd(dg, t, w, b=None):
    lib = _lib.load()
    N, C2, H, W = t.shape
    Cc = C2 // 2
    assert dg.is_contiguous() and t.is_contiguous()
    dt = torch.empty_like(t)
    dw = torch.empty(C2, 1, 3, 3, dtype=torch.float32, device=t.device)
    db = torch.empty(C2, dtype=torch.float32, device=t.device) if b is not None else None
    ws = workspace(lib.tdr_dwsg_ws_floats(N, Cc, H, W), t.device)
    check(lib.tdr_dwgelu_bwd(dg.data_ptr(), t.data_ptr(), w.data_ptr(), _p(b), N, Cc, H, W, dt.data_ptr(), dw.data_ptr(),
                             _p(db), ws.data_ptr(), _stream()), 'tdr_dwgelu_bwd')
    return dt, dw, db


def dwconv_fwd(t, w, b=None):
    """plain depthwise 3x3, pad 1"""
    N, Pn, H, W = t.shape
    assert t.is_contiguous()
    out = torch.empty_like(t)
    check(_lib.load().tdr_dwconv_fwd(t.data_ptr(), w.data_ptr(), _p(b), N, Pn, H, W, out.data_ptr(), _stream()), 'tdr_dwconv_fwd')
    return out


def dwconv_bwd(dout, t, w, want_db=False):
    lib = _lib.load()
    N, Pn, H, W = t.shape
    assert dout.is_contiguous() and t.is_contiguous()
    dt = torch.empty_like(t)
    dw = torch.empty(Pn, 1, 3, 3, dtype=torch.float32, device=t.device)
    db = torch.empty(Pn, dtype=torch.float32, device=t.device) if want_db else None
    ws = workspace(lib.tdr_dwsg_ws_floats(N, Pn // 2, H, W), t.device)
    check(lib.tdr_dwconv_bwd(dout.data_ptr(), t.data_ptr(), w.data_ptr(), N, Pn, H, W, dt.data_ptr(), dw.data_ptr(), _p(db),
                             ws.data_ptr(), _stream()), 'tdr_dwconv_bwd')
    return dt, dw, db


def row_sumsq(x, rows):
    """x [N, >=rows, H, W] (dense per image) -> [N, rows] sums of squares over H*W of the first `rows` channels"""
    N, _, H, W = x.shape
    out = torch.empty(N, rows, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_row_sumsq(x.data_ptr(), _dense_nchw(x), N, rows, H * W, out.data_ptr(), _stream()), 'tdr_row_sumsq')
    return out


def pack_f32packed_to_bx3(Wt):
    """Wt [B, Kp, Mp]: per-image 1x1 weights in the fp32 packed layout (Wt[b][cin][m] = W_b[m][cin], what tdr_mdta_* emit)
    -> (PackedWeights in the split-bf16 fragment layout, floats per image) for conv_forward(..., wp_ns=...)"""
    lib = _lib.load()
    B, Kp, Mp = Wt.shape
    assert Wt.is_contiguous()
    per_b = lib.tdr_packed_weight_bytes_bx3(Mp, Kp, 1) // 4
    buf = torch.empty(B * per_b, dtype=torch.float32, device=Wt.device)
    # mode 1 reads w[c * Cin + m]: with "Cin" = Mp (row length) and "Cout" = Kp this is exactly Wt[c][m]
    check(lib.tdr_pack_weights_bx3_batch(Wt.data_ptr(), Kp * Mp, B, Kp, Mp, 1, 1, buf.data_ptr(), _stream()),
          'tdr_pack_weights_bx3_batch')
    return PackedWeights(buf, FMT_BX3), per_b


def mdta_pad(Cc):
    return (Cc + 31) // 32 * 32


def mdta_softmax(G, ss, temp, heads):
    """G [N,C,C], ss [N,2C], temp [heads,1,1] -> (A, AT) [N,Cp,Cp] packed 1x1 weights (see include/tdr.h)"""
    N, Cc = G.shape[0], G.shape[-1]
    Cp = mdta_pad(Cc)
    A = torch.empty(N, Cp, Cp, dtype=torch.float32, device=G.device)
    AT = torch.empty_like(A)
    check(_lib.load().tdr_mdta_softmax(G.data_ptr(), ss.data_ptr(), temp.data_ptr(), N, Cc, heads, A.data_ptr(), AT.data_ptr(),
                                       _stream()), 'tdr_mdta_softmax')
    return A, AT


def mdta_bwd(G, ss, temp, A, dA, heads):
    """-> (W [N,Wp,Wp] packed weights of d[q;k] = W [q;k], dtemp [heads,1,1])"""
    N, Cc = G.shape[0], G.shape[-1]
    Wp = mdta_pad(2 * Cc)
    W = torch.empty(N, Wp, Wp, dtype=torch.float32, device=G.device)
    dtemp = torch.empty(heads, 1, 1, dtype=torch.float32, device=G.device)
    ws = torch.empty(N * heads, dtype=torch.float32, device=G.device)
    check(_lib.load().tdr_mdta_bwd(G.data_ptr(), ss.data_ptr(), temp.data_ptr(), A.data_ptr(), dA.data_ptr(), N, Cc, heads,
                                   W.data_ptr(), dtemp.data_ptr(), ws.data_ptr(), _stream()), 'tdr_mdta_bwd')
    return W, dtemp


def axpby_dev(a, alpha, b=None):
    """a * alpha[0] + b with alpha a device scalar"""
    assert a.is_contiguous() and (b is None or b.is_contiguous())
    out = torch.empty_like(a)
    check(_lib.load().tdr_axpby_dev(a.data_ptr(), alpha.data_ptr(), _p(b), a.numel(), out.data_ptr(), _stream()), 'tdr_axpby_dev')
    return out


def dot(a, b):
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = workspace(512, a.device, 'dot')
    check(_lib.load().tdr_dot(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), ws.data_ptr(), _stream()), 'tdr_dot')
    return out


def pixel_shuffle2(x):
    N, C4, H, W = x.shape
    assert x.is_contiguous() and C4 % 4 == 0
    out = torch.empty(N, C4 // 4, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_pixel_shuffle2(x.data_ptr(), N, C4 // 4, H, W, out.data_ptr(), _stream()), 'tdr_pixel_shuffle2')
    return out


# ------------------------------------------------------------------ stage-A mapper glue (main_train_i2t_mapping.py:40-81)
def leaky_relu_fwd(x, slope=0.01):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    check(_lib.load().tdr_leaky_relu_fwd(x.data_ptr(), x.numel(), float(slope), y.data_ptr(), _stream()), 'tdr_leaky_relu_fwd')
    return y


def leaky_relu_bwd(go, y, slope=0.01):
    assert go.is_contiguous() and y.is_contiguous()
    gx = torch.empty_like(go)
    check(_lib.load().tdr_leaky_relu_bwd(go.data_ptr(), y.data_ptr(), go.numel(), float(slope), gx.data_ptr(), _stream()),
          'tdr_leaky_relu_bwd')
    return gx


def gather_col(tok, col=0):
    """tok [B, D, LD/32, 32] -> [1, D, 1, 32]: column `col` of every image as pixel b (zero-padded to 32)"""
    B, D = tok.shape[0], tok.shape[1]
    LD = tok.shape[2] * tok.shape[3]
    assert tok.is_contiguous()
    out = torch.empty(1, D, 1, 32, dtype=torch.float32, device=tok.device)
    check(_lib.load().tdr_gather_col(tok.data_ptr(), B, D, LD, col, out.data_ptr(), _stream()), 'tdr_gather_col')
    return out


def mapper_combine(cls_out, patch_out, T, out, word):
    """out[:, word] = cls_out[d][b] + mean over the T patch columns of patch_out"""
    B, D = patch_out.shape[0], patch_out.shape[1]
    LD = patch_out.shape[2] * patch_out.shape[3]
    check(_lib.load().tdr_mapper_combine(cls_out.data_ptr(), patch_out.data_ptr(), B, D, LD, T, out.shape[1], word, out.data_ptr(),
                                         _stream()), 'tdr_mapper_combine')


def mapper_combine_bwd(go, LD, T, word):
    B, words, D = go.shape
    assert go.is_contiguous()
    dcls = torch.empty(1, D, 1, 32, dtype=torch.float32, device=go.device)
    dpatch = torch.empty(B, D, LD // 32, 32, dtype=torch.float32, device=go.device)
    check(_lib.load().tdr_mapper_combine_bwd(go.data_ptr(), B, D, LD, T, words, word, dcls.data_ptr(), dpatch.data_ptr(),
                                             _stream()), 'tdr_mapper_combine_bwd')
    return dcls, dpatch


def transpose_pad(src, LDd=None):
    """src [B, R, C] -> [B, C, LDd] with dst[b][c][r] = src[b][r][c], zero for r >= R"""
    B, R, Cc = src.shape
    assert src.is_contiguous()
    LDd = R if LDd is None else LDd
    out = torch.empty(B, Cc, LDd, dtype=torch.float32, device=src.device)
    check(_lib.load().tdr_transpose_pad(src.data_ptr(), B, R, Cc, LDd, out.data_ptr(), _stream()), 'tdr_transpose_pad')
    return out


def cross_attention_fwd(q, k, v, heads, scale, Tq, Tk):
    """q [B,C,LDq/32,32] (Tq valid columns), k/v [B,C,LDk/32,32] (Tk valid) -> (out like q, lse [B,heads,LDq])"""
    B, Cc = q.shape[0], q.shape[1]
    LDq, LDk = q.shape[2] * q.shape[3], k.shape[2] * k.shape[3]
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    out = torch.empty_like(q)
    lse = torch.empty(B, heads, LDq, dtype=torch.float32, device=q.device)
    check(_lib.load().tdr_cross_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), B, Cc, heads, Tq, LDq, Tk, LDk,
                                              float(scale), out.data_ptr(), lse.data_ptr(), _stream()), 'tdr_cross_attention_fwd')
    return out, lse


def cross_attention_bwd(q, k, v, out, dout, lse, heads, scale, Tq, Tk, need_dq=True):
    B, Cc = q.shape[0], q.shape[1]
    LDq, LDk = q.shape[2] * q.shape[3], k.shape[2] * k.shape[3]
    assert dout.is_contiguous() and out.is_contiguous()
    dq, dk, dv = (torch.empty_like(q) if need_dq else None), torch.empty_like(k), torch.empty_like(v)
    ws = workspace(B * heads * LDq, q.device, 'xattn')
    check(_lib.load().tdr_cross_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                              lse.data_ptr(), B, Cc, heads, Tq, LDq, Tk, LDk, float(scale), _p(dq),
                                              dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), _stream()), 'tdr_cross_attention_bwd')
    return dq, dk, dv


# ---- grouped Mapper glue (csrc/tdr_i2t.hip): tensors [G, C, H, W] with G = words, H*W = the tokens of all images
def group_ln_act_fwd(z, w, b, eps, slope):
    """nn.LayerNorm(C, eps) + nn.LeakyReLU(slope) with per-word parameters w, b [G, C] -> (y, mu [G, P], rstd [G, P])"""
    G, Cc, H, W = z.shape
    assert z.is_contiguous() and w.shape == (G, Cc) and b.shape == (G, Cc) and w.is_contiguous() and b.is_contiguous()
    y = torch.empty_like(z)
    mu = torch.empty(G, H * W, dtype=torch.float32, device=z.device)
    rs = torch.empty_like(mu)
    check(_lib.load().tdr_group_ln_act_fwd(z.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), float(slope), G, Cc, H * W, y.data_ptr(),
                                           mu.data_ptr(), rs.data_ptr(), _stream()), 'tdr_group_ln_act_fwd')
    return y, mu, rs


def group_ln_act_bwd(dy, y, z, mu, rs, w, slope):
    """-> (dz [G,C,H,W], gw [G,C], gb [G,C], gs [G,C] = sum over pixels of dz)"""
    G, Cc, H, W = z.shape
    assert dy.is_contiguous() and y.is_contiguous() and z.is_contiguous()
    lib = _lib.load()
    dz = torch.empty_like(z)
    gw = torch.empty(G, Cc, dtype=torch.float32, device=z.device)
    gb, gs = torch.empty_like(gw), torch.empty_like(gw)
    ws = workspace(lib.tdr_group_ln_ws_floats(G, Cc, H * W), z.device, 'gln')
    check(lib.tdr_group_ln_act_bwd(dy.data_ptr(), y.data_ptr(), z.data_ptr(), mu.data_ptr(), rs.data_ptr(), w.data_ptr(), float(slope), G, Cc,
                                   H * W, dz.data_ptr(), gw.data_ptr(), gb.data_ptr(), gs.data_ptr(), ws.data_ptr(), _stream()),
          'tdr_group_ln_act_bwd')
    return dz, gw, gb, gs


def mapper_combine_all(cls_out, patch_out, B, LD, T):
    """cls_out [G, D, 1, 32], patch_out [G, D, B*LD/32, 32] -> [B, G, D]"""
    G, D = patch_out.shape[0], patch_out.shape[1]
    out = torch.empty(B, G, D, dtype=torch.float32, device=patch_out.device)
    check(_lib.load().tdr_mapper_combine_all(cls_out.data_ptr(), patch_out.data_ptr(), B, G, D, LD, T, out.data_ptr(), _stream()),
          'tdr_mapper_combine_all')
    return out


def mapper_combine_all_bwd(go, LD, T):
    B, G, D = go.shape
    assert go.is_contiguous()
    dcls = torch.empty(G, D, 1, 32, dtype=torch.float32, device=go.device)
    dpatch = torch.empty(G, D, B * LD // 32, 32, dtype=torch.float32, device=go.device)
    gsum = torch.empty(G, D, dtype=torch.float32, device=go.device)
    check(_lib.load().tdr_mapper_combine_all_bwd(go.data_ptr(), B, G, D, LD, T, dcls.data_ptr(), dpatch.data_ptr(), gsum.data_ptr(), _stream()),
          'tdr_mapper_combine_all_bwd')
    return dcls, dpatch, gsum


def gather_col_flat(tok, B, col=0):
    """batch-flattened tokens [1, D, B*LD/32, 32] -> [1, D, 1, 32]: column `col` of every image as pixel b"""
    D = tok.shape[1]
    LD = tok.shape[2] * tok.shape[3] // B
    assert tok.is_contiguous()
    out = torch.empty(1, D, 1, 32, dtype=torch.float32, device=tok.device)
    check(_lib.load().tdr_gather_col_strided(tok.data_ptr(), B, D, LD, B * LD, col, out.data_ptr(), _stream()), 'tdr_gather_col_strided')
    return out


def splitk_finish(part, bias=None, scale=None, res=None, relu=0):
    """part [S, C, H, W] (K-chunk partial sums of a 1x1 convolution) -> [1, C, H, W] = act((sum_s part + bias) * scale + res)"""
    S, Cc, H, W = part.shape
    assert part.is_contiguous() and (res is None or (res.is_contiguous() and res.numel() == Cc * H * W))
    out = torch.empty(1, Cc, H, W, dtype=torch.float32, device=part.device)
    check(_lib.load().tdr_splitk_finish(part.data_ptr(), S, Cc, H * W, _p(bias), _p(scale), _p(res), int(relu) if not isinstance(relu, bool) else int(relu),
                                        out.data_ptr(), _stream()), 'tdr_splitk_finish')
    return out


# ---- stage-A train step glue (csrc/tdr_i2t.hip; main_train_i2t_mapping.py:704-760)
def _i32(t):
    assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous()
    return t.data_ptr()


def text_inject_fwd(ids, tok_emb, pos_emb, inj, idx):
    """ids [B,S] int32, tok_emb [V,D], pos_emb [S,D], inj [B,L,D], idx [B] int32 -> channel-major [B, D, LD/32, 32]"""
    B, S = ids.shape
    D, L = tok_emb.shape[1], inj.shape[1]
    LD = token_ld(S - 1)
    out = torch.empty(B, D, LD // 32, 32, dtype=torch.float32, device=inj.device)
    check(_lib.load().tdr_text_inject_fwd(_i32(ids), tok_emb.data_ptr(), pos_emb.data_ptr(), inj.contiguous().data_ptr(), _i32(idx),
                                          B, S, D, L, LD, out.data_ptr(), _stream()), 'tdr_text_inject_fwd')
    return out


def text_inject_bwd(dnew, idx, S, L):
    B, D = dnew.shape[0], dnew.shape[1]
    LD = dnew.shape[2] * dnew.shape[3]
    dinj = torch.empty(B, L, D, dtype=torch.float32, device=dnew.device)
    check(_lib.load().tdr_text_inject_bwd(dnew.data_ptr(), _i32(idx), B, S, D, L, LD, dinj.data_ptr(), _stream()), 'tdr_text_inject_bwd')
    return dinj


def add_noise(x, noise, t, alphas_cumprod):
    assert x.is_contiguous() and noise.is_contiguous() and x.shape == noise.shape
    out = torch.empty_like(x)
    check(_lib.load().tdr_add_noise(x.data_ptr(), noise.data_ptr(), _i32(t), alphas_cumprod.data_ptr(), x.shape[0],
                                    x.numel() // x.shape[0], out.data_ptr(), _stream()), 'tdr_add_noise')
    return out


def pool_time(x, t, f):
    B, Cc, H, W = x.shape
    assert x.is_contiguous()
    out = torch.empty(B, Cc + 4, H // f, W // f, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_pool_time(x.data_ptr(), _i32(t), B, Cc, H, W, f, out.data_ptr(), _stream()), 'tdr_pool_time')
    return out


def upsample_nearest_add_(dst, src, f, accumulate=True):
    B, Cc, H, W = dst.shape
    assert dst.is_contiguous() and src.is_contiguous() and src.shape == (B, Cc, H // f, W // f)
    check(_lib.load().tdr_upsample_nearest_add(src.data_ptr(), B * Cc, H, W, f, 1 if accumulate else 0, dst.data_ptr(), _stream()),
          'tdr_upsample_nearest_add')
    return dst


def pool_sum(src, f):
    B, Cc, H, W = src.shape
    assert src.is_contiguous()
    out = torch.empty(B, Cc, H // f, W // f, dtype=torch.float32, device=src.device)
    check(_lib.load().tdr_pool_sum(src.data_ptr(), B * Cc, H, W, f, out.data_ptr(), _stream()), 'tdr_pool_sum')
    return out


# ---------------------------------------------------------------------------
# PromptIR-ref PromptGenBlock pieces (csrc/tdr_prompt.hip)
# ---------------------------------------------------------------------------
def plane_mean(x):
    """x [N,C,H,W] (dense NCHW view) -> [N,C] mean over H*W."""
    N, Cc, H, W = x.shape
    out = torch.empty(N, Cc, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_plane_mean(x.data_ptr(), _dense_nchw(x), N, Cc, H * W, out.data_ptr(), _stream()), 'tdr_plane_mean')
    return out


def plane_add_(x, v, scale):
    """x[n,c,:,:] += v[n,c] * scale, in place."""
    N, Cc, H, W = x.shape
    check(_lib.load().tdr_plane_add(x.data_ptr(), _dense_nchw(x), v.data_ptr(), float(scale), N, Cc, H * W, _stream()), 'tdr_plane_add')
    return x


def prompt_weights_fwd(emb, W, b):
    N, Cc = emb.shape
    L = W.shape[0]
    w = torch.empty(N, L, dtype=torch.float32, device=emb.device)
    check(_lib.load().tdr_prompt_weights_fwd(emb.data_ptr(), W.data_ptr(), _p(b), N, Cc, L, w.data_ptr(), _stream()),
          'tdr_prompt_weights_fwd')
    return w


def prompt_weights_bwd(emb, W, w, dw, want_db=True):
    N, Cc = emb.shape
    L = W.shape[0]
    dW = torch.empty_like(W)
    db = torch.empty(L, dtype=torch.float32, device=emb.device) if want_db else None
    demb = torch.empty_like(emb)
    check(_lib.load().tdr_prompt_weights_bwd(emb.data_ptr(), W.data_ptr(), w.data_ptr(), dw.data_ptr(), N, Cc, L, dW.data_ptr(),
                                             _p(db), demb.data_ptr(), _stream()), 'tdr_prompt_weights_bwd')
    return dW, db, demb


def prompt_mix_fwd(w, P):
    """w [N,L], P [L,C,H,W] -> [N,C,H,W] = sum_k w[n,k] P[k]."""
    N, L = w.shape
    out = torch.empty(N, *P.shape[1:], dtype=torch.float32, device=P.device)
    check(_lib.load().tdr_prompt_mix_fwd(w.data_ptr(), P.data_ptr(), N, L, P[0].numel(), out.data_ptr(), _stream()), 'tdr_prompt_mix_fwd')
    return out


def prompt_mix_bwd(w, P, d):
    lib = _lib.load()
    N, L = w.shape
    assert d.is_contiguous() and P.is_contiguous()
    dP = torch.empty_like(P)
    dw = torch.empty_like(w)
    ws = workspace(lib.tdr_prompt_mix_bwd_ws_floats(N, L), P.device)
    check(lib.tdr_prompt_mix_bwd(w.data_ptr(), P.data_ptr(), d.data_ptr(), N, L, P[0].numel(), dP.data_ptr(), dw.data_ptr(),
                                 ws.data_ptr(), _stream()), 'tdr_prompt_mix_bwd')
    return dP, dw


def resize_bilinear_bwd(dd, Hs, Ws):
    B, Cc, Hd, Wd = dd.shape
    assert dd.is_contiguous()
    ds = torch.empty(B, Cc, Hs, Ws, dtype=torch.float32, device=dd.device)
    check(_lib.load().tdr_resize_bilinear_bwd(dd.data_ptr(), B * Cc, Hs, Ws, Hd, Wd, ds.data_ptr(), _stream()), 'tdr_resize_bilinear_bwd')
    return ds


def crop_augment(src, top, left, mode, patch, noise=None, sigma=None):
    """src [N,C,H,W] -> [N,C,patch,patch]: per-sample crop at (top, left) + augmentation mode 0-7 (+ noise * sigma).
    top / left / mode: int32 device tensors [N] (or None); noise [N,C,patch,patch], sigma [N] float32 (or None)."""
    N, Cc, H, W = src.shape
    out = torch.empty(N, Cc, patch, patch, dtype=torch.float32, device=src.device)
    for t in (top, left, mode):
        assert t is None or (t.dtype == torch.int32 and t.is_cuda and t.numel() == N)
    check(_lib.load().tdr_crop_augment(src.data_ptr(), _dense_nchw(src), N, Cc, H, W, _p(top), _p(left), _p(mode), _p(noise), _p(sigma),
                                       int(patch), out.data_ptr(), _stream()), 'tdr_crop_augment')
    return out


def ssim3d(img1, img2, max_value):
    """img1 / img2 [H,W,C] float32 device tensors (C <= 4) -> 1-element tensor: mean SSIM over the volume with the 11^3
    Gaussian window and replicate borders (metrics/psnr_ssim.py:131-176)."""
    assert img1.shape == img2.shape and img1.dim() == 3 and img1.is_contiguous() and img2.is_contiguous()
    H, W, Cc = img1.shape
    lib = _lib.load()
    ws = workspace(lib.tdr_ssim3d_ws_floats(H, W), img1.device, 'ssim')
    out = torch.empty(1, dtype=torch.float32, device=img1.device)
    check(lib.tdr_ssim3d(img1.data_ptr(), img2.data_ptr(), H, W, Cc, float(max_value), ws.data_ptr(), out.data_ptr(), _stream()),
          'tdr_ssim3d')
    return out


# ---------------------------------------------------------------------------
# DRSformer-ref pieces (csrc/tdr_mdta.hip: top-k sparse attention; csrc/tdr_dwk.hip: grouped depthwise convs)
# ---------------------------------------------------------------------------
def ssim_y64(img1, img2):
    """float64 SSIM of two [H, W] float32 planes (metrics/psnr_ssim.py:184-222); returns a python float"""
    lib = _lib.load()
    H, W = img1.shape
    assert img1.is_contiguous() and img2.is_contiguous() and img2.shape == img1.shape
    ws = torch.empty(lib.tdr_ssim_y64_ws_doubles(H, W), dtype=torch.float64, device=img1.device)
    out = torch.empty(1, dtype=torch.float64, device=img1.device)
    check(lib.tdr_ssim_y64(img1.data_ptr(), img2.data_ptr(), H, W, ws.data_ptr(), out.data_ptr(), _stream()), 'tdr_ssim_y64')
    return float(out.item())


def local_avgpool(x, k1, k2):
    """TLSC box mean of x [N, C, H, W] with replicate padding back to H x W (nafnet_local_arch.py:10-75)"""
    lib = _lib.load()
    N, Cc, H, W = x.shape
    assert x.is_contiguous()
    out = torch.empty_like(x)
    ws = workspace(lib.tdr_local_avgpool_ws_floats(N * Cc, H, W, k1), x.device, 'tlsc')
    check(lib.tdr_local_avgpool(x.data_ptr(), N * Cc, H, W, k1, k2, ws.data_ptr(), out.data_ptr(), _stream()), 'tdr_local_avgpool')
    return out


def tksa_ks(c):
    """the four top-k sizes of a head with c channels, as the reference computes them (python int() of float expressions,
    network_drsformer_guided_arch.py:293-306)"""
    return (C.c_int * 4)(int(c / 2), int(c * 2 / 3), int(c * 3 / 4), int(c * 4 / 5))


def tksa_softmax(G, ss, temp, am, heads):
    """G [N,C,C], ss [N,2C], am [4] (attn1..4) -> (A, AT) packed like mdta_softmax, A = sum_m am[m] softmax(topk_m(logits))"""
    N, Cc = G.shape[0], G.shape[-1]
    Cp = mdta_pad(Cc)
    A = torch.empty(N, Cp, Cp, dtype=torch.float32, device=G.device)
    AT = torch.empty_like(A)
    check(_lib.load().tdr_tksa_softmax(G.data_ptr(), ss.data_ptr(), temp.data_ptr(), am.data_ptr(), tksa_ks(Cc // heads), N, Cc, heads,
                                       A.data_ptr(), AT.data_ptr(), _stream()), 'tdr_tksa_softmax')
    return A, AT


def tksa_bwd(G, ss, temp, am, dA, heads):
    """-> (W [N,Wp,Wp], dtemp [heads,1,1], dam [4])"""
    N, Cc = G.shape[0], G.shape[-1]
    Wp = mdta_pad(2 * Cc)
    W = torch.empty(N, Wp, Wp, dtype=torch.float32, device=G.device)
    dtemp = torch.empty(heads, 1, 1, dtype=torch.float32, device=G.device)
    dam = torch.empty(4, dtype=torch.float32, device=G.device)
    ws = torch.empty(5 * N * heads, dtype=torch.float32, device=G.device)
    check(_lib.load().tdr_tksa_bwd(G.data_ptr(), ss.data_ptr(), temp.data_ptr(), am.data_ptr(), tksa_ks(Cc // heads), dA.data_ptr(), N, Cc,
                                   heads, W.data_ptr(), dtemp.data_ptr(), dam.data_ptr(), ws.data_ptr(), _stream()), 'tdr_tksa_bwd')
    return W, dtemp, dam


def _dw3_plain(x, w, dil):
    Cout, mult, Kk, _ = w.shape
    return Kk == 3 and mult == 1 and dil == 1 and Cout % 2 == 0 and x.shape[-1] % 4 == 0 and x.is_contiguous() and \
        not DWK_GENERIC


def _dw3_pair(x, w, dil):
    Cout, mult, Kk, _ = w.shape
    return Kk == 3 and mult == 2 and dil == 1 and x.shape[-1] % 4 == 0 and x.is_contiguous() and \
        not DWK_GENERIC


def _DW_TWO_PASS():
    return os.environ.get('TDR_DWSG_TWO_PASS', '0') == '1'


def dwk_fwd(x, w, b=None, relu=False, dil=1, out=None):
    """grouped depthwise-like conv: w [Cout, mult, K, K] (mult 1 | 2), stride 1, pad dil * (K // 2), optional bias / fused ReLU.
    out: optional destination [N, Cout, H, W], dense per image -- e.g. a channel slice of a concatenation buffer"""
    N, Cin, H, W = x.shape
    Cout, mult, Kk, _ = w.shape
    assert Cin == Cout * mult and w.is_contiguous()
    if isinstance(out, tuple):    # plain depthwise 3x3 only: planes [0, Cout/2) -> out[0], planes [Cout/2, Cout) -> out[1] (channel slices)
        ya, yb = out
        assert _dw3_plain(x, w, dil) and ya.shape == yb.shape == (N, Cout // 2, H, W) and _dense_nchw(ya) == _dense_nchw(yb)
        check(_lib.load().tdr_dwconv_halves_fwd(x.data_ptr(), w.data_ptr(), _p(b), N, Cout, H, W, 1 if relu else 0, ya.data_ptr(),
                                                yb.data_ptr(), _dense_nchw(ya), _stream()), 'tdr_dwconv_halves_fwd')
        return out
    y = torch.empty(N, Cout, H, W, dtype=torch.float32, device=x.device) if out is None else out
    assert y.shape == (N, Cout, H, W)
    if _dw3_plain(x, w, dil) and y.is_contiguous():     # plain depthwise 3x3: the register-window stencil of tdr_dwsg.hip (plane pairs c, c + Cout/2)
        check(_lib.load().tdr_dwconv_act_fwd(x.data_ptr(), w.data_ptr(), _p(b), N, Cout, H, W, 1 if relu else 0, y.data_ptr(), _stream()),
              'tdr_dwconv_act_fwd')
        return y
    if _dw3_pair(x, w, dil):      # two inputs per output, 3x3: the same stencil with (2c, 2c+1) plane pairs
        check(_lib.load().tdr_dwpair_fwd(x.data_ptr(), w.data_ptr(), _p(b), N, Cout, H, W, 1 if relu else 0, y.data_ptr(), _dense_nchw(y),
                                         _stream()), 'tdr_dwpair_fwd')
        return y
    check(_lib.load().tdr_dwk_fwd(x.data_ptr(), _dense_nchw(x), w.data_ptr(), _p(b), N, Cout, mult, H, W, Kk, int(dil), 1 if relu else 0,
                                  y.data_ptr(), _dense_nchw(y), _stream()), 'tdr_dwk_fwd')
    return y


def dwk_bwd(dy, y_act, x, w, want_db=False, dil=1, dx_out=None, dw_out=None, db_out=None, accumulate=False):
    """-> (dx, dw, db); y_act: the forward output when a ReLU was fused (its mask), else None.  dy / y_act may be (first half, second
    half) tuples of channel slices for the plain depthwise 3x3 (see dwk_fwd); dx_out / dw_out / db_out: optional destinations (dx_out
    dense per image, e.g. a channel slice); accumulate: dx_out += instead of = (inside the one-pass 5x5 kernel, else a separate add)"""
    N, Cin, H, W = x.shape
    Cout, mult, Kk, _ = w.shape
    dx = torch.empty(N, Cin, H, W, dtype=torch.float32, device=x.device) if dx_out is None else dx_out
    dw = torch.empty_like(w) if dw_out is None else dw_out
    db = (torch.empty(Cout, dtype=torch.float32, device=x.device) if db_out is None else db_out) if want_db else None
    assert dx.shape == (N, Cin, H, W) and dw.is_contiguous() and dw.shape == w.shape
    lib = _lib.load()
    if isinstance(dy, tuple):
        da, db_ = dy
        aa, ab = y_act if y_act is not None else (None, None)
        assert _dw3_plain(x, w, dil) and dx.is_contiguous() and _dense_nchw(da) == _dense_nchw(db_)
        assert aa is None or _dense_nchw(aa) == _dense_nchw(ab)
        ws = workspace(lib.tdr_dwsg_ws_floats(N, Cout // 2, H, W), x.device)
        check(lib.tdr_dwconv_halves_bwd(da.data_ptr(), db_.data_ptr(), _dense_nchw(da), _p(aa), _p(ab), _dense_nchw(aa) if aa is not None else 0,
                                        x.data_ptr(), w.data_ptr(), N, Cout, H, W, dx.data_ptr(), dw.data_ptr(), _p(db), ws.data_ptr(),
                                        _stream()), 'tdr_dwconv_halves_bwd')
        return dx, dw, db
    if _dw3_plain(x, w, dil) and dy.is_contiguous() and (y_act is None or y_act.is_contiguous()) and not _DW_TWO_PASS() and dx.is_contiguous():
        ws = workspace(lib.tdr_dwsg_ws_floats(N, Cout // 2, H, W), x.device)
        check(lib.tdr_dwconv_act_bwd(dy.data_ptr(), _p(y_act), x.data_ptr(), w.data_ptr(), N, Cout, H, W, dx.data_ptr(), dw.data_ptr(),
                                     _p(db), ws.data_ptr(), _stream()), 'tdr_dwconv_act_bwd')
        return dx, dw, db
    if _dw3_pair(x, w, dil) and W <= 1024 and dx.is_contiguous():
        ws = workspace(lib.tdr_dwsg_ws_floats(N, Cout, H, W), x.device)
        check(lib.tdr_dwpair_bwd(dy.data_ptr(), _dense_nchw(dy), _p(y_act), _dense_nchw(y_act) if y_act is not None else 0, x.data_ptr(),
                                 w.data_ptr(), N, Cout, H, W, dx.data_ptr(), dw.data_ptr(), _p(db), ws.data_ptr(), _stream()),
              'tdr_dwpair_bwd')
        return dx, dw, db
    ws = workspace(lib.tdr_dwk_bwd_ws_floats(N, Cout, mult, H, W, Kk), x.device, 'dwk')
    y_ns = _dense_nchw(y_act) if y_act is not None else 0
    acc_in_kernel = accumulate and lib.tdr_dwk_bwd_can_accumulate(W, Kk, int(dil), _dense_nchw(dy), y_ns, _dense_nchw(x), _dense_nchw(dx)) != 0
    tgt = dx if (not accumulate or acc_in_kernel) else torch.empty(N, Cin, H, W, dtype=torch.float32, device=x.device)
    check(lib.tdr_dwk_bwd_acc(dy.data_ptr(), _dense_nchw(dy), _p(y_act), y_ns, x.data_ptr(), _dense_nchw(x), w.data_ptr(), N, Cout, mult,
                              H, W, Kk, int(dil), tgt.data_ptr(), _dense_nchw(tgt), 1 if acc_in_kernel else 0, dw.data_ptr(), _p(db),
                              ws.data_ptr(), _stream()), 'tdr_dwk_bwd')
    if tgt is not dx:
        add_(dx, tgt)
    return dx, dw, db


def avgpool3(x, adjoint=False):
    """nn.AvgPool2d(3, 1, 1, count_include_pad=False) (adjoint: its backward)"""
    N, Cc, H, W = x.shape
    assert x.is_contiguous()
    out = torch.empty_like(x)
    check(_lib.load().tdr_avgpool3(x.data_ptr(), N * Cc, H, W, 1 if adjoint else 0, out.data_ptr(), _stream()), 'tdr_avgpool3')
    return out


def linear_small_fwd(x, W, b=None, relu=False):
    N, Cin = x.shape
    y = torch.empty(N, W.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_linear_small_fwd(x.data_ptr(), W.data_ptr(), _p(b), N, Cin, W.shape[0], 1 if relu else 0, y.data_ptr(), _stream()),
          'tdr_linear_small_fwd')
    return y


def linear_small_bwd(dy, y_act, x, W, want_db=True):
    N, Cin = x.shape
    dx, dW = torch.empty_like(x), torch.empty_like(W)
    db = torch.empty(W.shape[0], dtype=torch.float32, device=x.device) if want_db else None
    check(_lib.load().tdr_linear_small_bwd(dy.data_ptr(), _p(y_act), x.data_ptr(), W.data_ptr(), N, Cin, W.shape[0], dx.data_ptr(),
                                           dW.data_ptr(), _p(db), _stream()), 'tdr_linear_small_bwd')
    return dx, dW, db


def softmax_rows(x, dy=None):
    """x [rows, L]: softmax over L; with dy: backward, x being the softmax OUTPUT"""
    rows, L = x.shape
    out = torch.empty_like(x)
    check(_lib.load().tdr_softmax_rows(x.data_ptr(), _p(dy), rows, L, out.data_ptr(), _stream()), 'tdr_softmax_rows')
    return out


def scale_copy(src, w, w_stride, dst):
    """dst[n] = src[n] * w[n * w_stride] (w: a view into the [N, steps, ops] weight tensor starting at the wanted element)"""
    N = src.shape[0]
    length = src[0].numel()
    check(_lib.load().tdr_scale_copy(src.data_ptr(), _dense_nchw(src), w.data_ptr(), int(w_stride), N, length, dst.data_ptr(),
                                     _dense_nchw(dst), _stream()), 'tdr_scale_copy')
    return dst


def rows_dot(a, b, out, out_stride):
    """out[n * out_stride] = <a[n], b[n]>"""
    N = a.shape[0]
    ws = workspace(64 * N, a.device, 'rowsdot')
    check(_lib.load().tdr_rows_dot(a.data_ptr(), _dense_nchw(a), b.data_ptr(), _dense_nchw(b), N, a[0].numel(), out.data_ptr(),
                                   int(out_stride), ws.data_ptr(), _stream()), 'tdr_rows_dot')
    return out


def add_relu(a, b):
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    check(_lib.load().tdr_add_relu(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream()), 'tdr_add_relu')
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# un-guided SFNet (csrc/tdr_sfnet.hip; models/archs/sfnet_arch_utils.py:76-265 of the reference)
# ---------------------------------------------------------------------------------------------------------------------------------
def gelu_fwd(x, bias=None):
    """-> (z, y): z = x + bias[channel] (x itself without a bias), y = gelu(z) (exact erf).  x [N, C, H, W] dense."""
    N, Cc, H, W = x.shape
    y = torch.empty_like(x)
    z = x if bias is None else torch.empty_like(x)
    check(_lib.load().tdr_gelu_fwd(x.data_ptr(), _p(bias), Cc, H * W, z.data_ptr() if bias is not None else 0, y.data_ptr(), x.numel(),
                                   _stream()), 'tdr_gelu_fwd')
    return z, y


def gelu_bwd(dy, z):
    dz = torch.empty_like(z)
    check(_lib.load().tdr_gelu_bwd(dy.data_ptr(), z.data_ptr(), dz.data_ptr(), z.numel(), _stream()), 'tdr_gelu_bwd')
    return dz


def subsample2(x):
    """F.interpolate(scale_factor=0.5), nearest: x[..., ::2, ::2]"""
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, H // 2, W // 2, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_subsample2(x.data_ptr(), N * Cc, H, W, y.data_ptr(), _stream()), 'tdr_subsample2')
    return y


def instnorm_fwd(x, w, b, eps=1e-5):
    N, Cc, H, W = x.shape
    y = torch.empty_like(x)
    mu = torch.empty(N * Cc, dtype=torch.float32, device=x.device)
    rs = torch.empty_like(mu)
    check(_lib.load().tdr_instnorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), N, Cc, H * W, y.data_ptr(), mu.data_ptr(),
                                       rs.data_ptr(), _stream()), 'tdr_instnorm_fwd')
    return y, mu, rs


def instnorm_bwd(dy, x, mu, rs, w):
    N, Cc, H, W = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty(Cc, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dw)
    ws = torch.empty(2 * N * Cc, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_instnorm_bwd(dy.data_ptr(), x.data_ptr(), mu.data_ptr(), rs.data_ptr(), w.data_ptr(), N, Cc, H * W, dx.data_ptr(),
                                       dw.data_ptr(), db.data_ptr(), ws.data_ptr(), _stream()), 'tdr_instnorm_bwd')
    return dx, dw, db


def region_affine_fwd(x, ph, pl, shift, q, out):
    """out = x * A + mean_region(x) * B with A = ph + shift, B = pl - A over the q x q equal blocks of the plane (Gap: q = 1, shift = 1,
    (fscale_h, fscale_d); Patch_ap: q = 2, shift = 0, (h, l)); x / out: dense-NCHW views (channel slices of bigger buffers are fine).
    Returns the region means [N, C, q * q] (kept for the backward pass)."""
    N, Cc, H, W = x.shape
    mean = torch.empty(N, Cc, q * q, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_region_affine_fwd(x.data_ptr(), _dense_nchw(x), ph.data_ptr(), pl.data_ptr(), float(shift), q, N, Cc, H, W,
                                            out.data_ptr(), _dense_nchw(out), mean.data_ptr(), _stream()), 'tdr_region_affine_fwd')
    return mean


def region_affine_bwd(dy, x, ph, pl, shift, mean, q, dx):
    """-> (dph, dpl) [C * q * q]; dx (a dense-NCHW view) receives the data gradient"""
    N, Cc, H, W = x.shape
    J = Cc * q * q
    dph = torch.empty(J, dtype=torch.float32, device=x.device)
    dpl = torch.empty_like(dph)
    ws = torch.empty(2 * N * J, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_region_affine_bwd(dy.data_ptr(), _dense_nchw(dy), x.data_ptr(), _dense_nchw(x), ph.data_ptr(), pl.data_ptr(),
                                            float(shift), mean.data_ptr(), q, N, Cc, H, W, dx.data_ptr(), _dense_nchw(dx), dph.data_ptr(),
                                            dpl.data_ptr(), ws.data_ptr(), _stream()), 'tdr_region_affine_bwd')
    return dph, dpl

def sf_region_split(x, q):
    """the q x q region planes of a dense-NCHW view as a contiguous [N, C q q, H / q, W / q] tensor (q = 1: a dense copy of a channel slice)"""
    N, Cc, H, W = x.shape
    out = torch.empty(N, Cc * q * q, H // q, W // q, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_sf_region_split(x.data_ptr(), _dense_nchw(x), q, N, Cc, H, W, out.data_ptr(), _stream()), 'tdr_sf_region_split')
    return out


def sf_local_affine(x, m, ph, pl, shift, q, out):
    """out = m * pl + (x - m) * (ph + shift) with m [N, C q q, H / q, W / q] the TLSC box-mean map of x's region planes (SFNet mode 'test')"""
    N, Cc, H, W = x.shape
    assert m.is_contiguous() and m.shape == (N, Cc * q * q, H // q, W // q)
    check(_lib.load().tdr_sf_local_affine(x.data_ptr(), _dense_nchw(x), m.data_ptr(), ph.data_ptr(), pl.data_ptr(), float(shift), q, N, Cc, H, W,
                                          out.data_ptr(), _dense_nchw(out), _stream()), 'tdr_sf_local_affine')
    return out


def sf_emerge(x, low):
    N, Cc, H, W = x.shape
    out = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_sf_emerge(x.data_ptr(), _dense_nchw(x), low.data_ptr(), N, Cc, H * W, out.data_ptr(), _stream()), 'tdr_sf_emerge')
    return out


def sf_softmax_mix(x, low, lh, ll):
    """per pixel: softmax over the 2c logits [lh ; ll], mix = (x - low) * a_high + low * a_low"""
    N, Cc, H, W = x.shape
    assert low.is_contiguous() and lh.is_contiguous() and ll.is_contiguous()
    mix = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_sf_softmax_mix(x.data_ptr(), _dense_nchw(x), low.data_ptr(), lh.data_ptr(), ll.data_ptr(), N, Cc, H * W, mix.data_ptr(),
                                         _stream()), 'tdr_sf_softmax_mix')
    return mix



def sf_dyn_vec_fwd(ap, P, pre, k, groups=8, training=True):
    """the pooled-vector pipeline of dynamic_filter + SFconv (one workgroup): -> (taps [N, G k k], ah [N, c], al [N, c], saved).
    training: P[pre + 'bn.running_mean' / 'running_var' / 'num_batches_tracked'] are moved in place (training-mode BatchNorm);
    not training (module.eval()): the running statistics normalise and nothing is updated."""
    N, c = ap.shape
    KK, GK = k * k, groups * k * k
    dd = P[pre + 'modulate.fc.weight'].shape[0]
    dev = ap.device
    taps = torch.empty(N, GK, dtype=torch.float32, device=dev)
    ah = torch.empty(N, c, dtype=torch.float32, device=dev)
    al = torch.empty_like(ah)
    xhat = torch.empty(N, GK, dtype=torch.float32, device=dev)
    rstd = torch.empty(GK, dtype=torch.float32, device=dev)
    z = torch.empty(N, dd, dtype=torch.float32, device=dev)
    att = torch.empty(N, 2 * c, dtype=torch.float32, device=dev)
    d = _lib.TdrSfDynVecDesc()
    d.N, d.c, d.GK, d.KK, d.d, d.eps, d.momentum = N, c, GK, KK, dd, 1e-5, 0.1
    d.ap, d.wconv, d.bn_w, d.bn_b = ap.data_ptr(), P[pre + 'conv.weight'].data_ptr(), P[pre + 'bn.weight'].data_ptr(), P[pre + 'bn.bias'].data_ptr()
    d.fc_w, d.fc_b = P[pre + 'modulate.fc.weight'].data_ptr(), P[pre + 'modulate.fc.bias'].data_ptr()
    d.f0_w, d.f0_b = P[pre + 'modulate.fcs.0.weight'].data_ptr(), P[pre + 'modulate.fcs.0.bias'].data_ptr()
    d.f1_w, d.f1_b = P[pre + 'modulate.fcs.1.weight'].data_ptr(), P[pre + 'modulate.fcs.1.bias'].data_ptr()
    d.run_mean, d.run_var = P[pre + 'bn.running_mean'].data_ptr(), P[pre + 'bn.running_var'].data_ptr()
    nbt = P.get(pre + 'bn.num_batches_tracked')
    d.nbt = nbt.data_ptr() if nbt is not None and nbt.is_cuda else 0
    d.use_running = 0 if training else 1
    d.taps, d.ah, d.al, d.xhat, d.rstd, d.z, d.att = (t.data_ptr() for t in (taps, ah, al, xhat, rstd, z, att))
    check(_lib.load().tdr_sf_dyn_vec_fwd(C.byref(d), _stream()), 'tdr_sf_dyn_vec_fwd')
    return taps, ah, al, (ap, taps, xhat, rstd, z, att)


def sf_dyn_vec_bwd(dtaps, dah, dal, P, pre, k, saved, groups=8):
    """-> (dap [N, c], {parameter name (without pre): gradient})"""
    ap, taps, xhat, rstd, z, att = saved
    N, c = ap.shape
    KK, GK = k * k, groups * k * k
    dd = z.shape[1]
    dev = ap.device
    dap = torch.empty(N, c, dtype=torch.float32, device=dev)
    G = {n: torch.empty_like(P[pre + n]) for n in ('conv.weight', 'bn.weight', 'bn.bias', 'modulate.fc.weight', 'modulate.fc.bias',
                                                   'modulate.fcs.0.weight', 'modulate.fcs.0.bias', 'modulate.fcs.1.weight', 'modulate.fcs.1.bias')}
    ws = torch.empty(N * (2 * c + dd + GK), dtype=torch.float32, device=dev)
    d = _lib.TdrSfDynVecBwdDesc()
    d.N, d.c, d.GK, d.KK, d.d = N, c, GK, KK, dd
    d.ap, d.wconv, d.bn_w, d.fc_w = ap.data_ptr(), P[pre + 'conv.weight'].data_ptr(), P[pre + 'bn.weight'].data_ptr(), P[pre + 'modulate.fc.weight'].data_ptr()
    d.f0_w, d.f1_w = P[pre + 'modulate.fcs.0.weight'].data_ptr(), P[pre + 'modulate.fcs.1.weight'].data_ptr()
    d.taps, d.xhat, d.rstd, d.z, d.att = taps.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), z.data_ptr(), att.data_ptr()
    d.dtaps, d.dah, d.dal, d.dap = dtaps.data_ptr(), dah.data_ptr(), dal.data_ptr(), dap.data_ptr()
    d.g_wconv, d.g_bn_w, d.g_bn_b = G['conv.weight'].data_ptr(), G['bn.weight'].data_ptr(), G['bn.bias'].data_ptr()
    d.g_fc_w, d.g_fc_b = G['modulate.fc.weight'].data_ptr(), G['modulate.fc.bias'].data_ptr()
    d.g_f0_w, d.g_f0_b = G['modulate.fcs.0.weight'].data_ptr(), G['modulate.fcs.0.bias'].data_ptr()
    d.g_f1_w, d.g_f1_b = G['modulate.fcs.1.weight'].data_ptr(), G['modulate.fcs.1.bias'].data_ptr()
    d.ws = ws.data_ptr()
    check(_lib.load().tdr_sf_dyn_vec_bwd(C.byref(d), _stream()), 'tdr_sf_dyn_vec_bwd')
    return dap, G


def sf_dynfilt_fwd(x, taps, ah, al, k, groups=8):
    """x: dense-NCHW view -> (low, mix) dense [N, C, H, W]"""
    N, Cc, H, W = x.shape
    low = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x.device)
    mix = torch.empty_like(low)
    check(_lib.load().tdr_sf_dynfilt_fwd(x.data_ptr(), _dense_nchw(x), taps.data_ptr(), ah.data_ptr(), al.data_ptr(), N, Cc, groups, H, W, k,
                                         low.data_ptr(), mix.data_ptr(), _stream()), 'tdr_sf_dynfilt_fwd')
    return low, mix


def sf_dynfilt_bwd_reduce(dmix, x, low, ah, al, k, groups=8):
    N, Cc, H, W = x.shape
    dah = torch.empty(N, Cc, dtype=torch.float32, device=x.device)
    dal = torch.empty_like(dah)
    dtaps = torch.empty(N, groups * k * k, dtype=torch.float32, device=x.device)
    check(_lib.load().tdr_sf_dynfilt_bwd_reduce(dmix.data_ptr(), x.data_ptr(), _dense_nchw(x), low.data_ptr(), ah.data_ptr(), al.data_ptr(), N,
                                                Cc, groups, H, W, k, dah.data_ptr(), dal.data_ptr(), dtaps.data_ptr(), _stream()),
          'tdr_sf_dynfilt_bwd_reduce')
    return dah, dal, dtaps


def sf_dynfilt_bwd_dx(dmix, taps, ah, al, dap, k, dx, groups=8):
    N, Cc, H, W = dmix.shape
    check(_lib.load().tdr_sf_dynfilt_bwd_dx(dmix.data_ptr(), taps.data_ptr(), ah.data_ptr(), al.data_ptr(), dap.data_ptr(), N, Cc, groups, H,
                                            W, k, dx.data_ptr(), _dense_nchw(dx), _stream()), 'tdr_sf_dynfilt_bwd_dx')
    return dx


def convt4_weight_to_3x3(w, b):
    """ConvTranspose2d(4, 2, 1) weight [Cin, Cout, 4, 4] (+ bias) -> the 3x3 / pad 1 convolution [4 Cout, Cin, 3, 3] whose PixelShuffle(2)
    is the transposed convolution (+ bias repeated per parity)"""
    Cin, Cout = w.shape[0], w.shape[1]
    w3 = torch.empty(4 * Cout, Cin, 3, 3, dtype=torch.float32, device=w.device)
    b4 = torch.empty(4 * Cout, dtype=torch.float32, device=w.device) if b is not None else None
    check(_lib.load().tdr_convt4_weight_to_3x3(w.data_ptr(), _p(b), Cin, Cout, w3.data_ptr(), _p(b4), _stream()), 'tdr_convt4_weight_to_3x3')
    return w3, b4


def convt4_grad_from_3x3(dw3, db4, Cin, Cout):
    dw = torch.empty(Cin, Cout, 4, 4, dtype=torch.float32, device=dw3.device)
    db = torch.empty(Cout, dtype=torch.float32, device=dw3.device) if db4 is not None else None
    check(_lib.load().tdr_convt4_grad_from_3x3(dw3.data_ptr(), _p(db4), Cin, Cout, dw.data_ptr(), _p(db), _stream()), 'tdr_convt4_grad_from_3x3')
    return dw, db

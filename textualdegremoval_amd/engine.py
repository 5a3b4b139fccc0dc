"""Hand-written forward/backward of the NAFNet-ref hot path on the HIP kernels.

Functional pieces (`*_fwd` returns (out, saved), `*_bwd` returns (dx, grads))
mirror the reference modules they replace and cite them; parameters are passed
as dicts keyed by the reference's state-dict names, gradients come back keyed
the same way.  No ATen arithmetic runs on the device here: every tensor op is a
call into libtdr_hip.so (torch only allocates and views memory).
"""
import os

import torch

from . import kernels as K
from .kernels import (EPI_GATEBWD, EPI_PSHUF, PACK_DGRAD_2X2S2, PACK_DGRAD_3X3S2, PACK_DGRAD_S1, PACK_FWD)

LN_EPS = 1e-6
# fused NAFBlock halves (csrc/tdr_nafblock.hip) where the shape allows (module switches -- tests and probes set them; none is an environment knob)
FUSE_TAIL = True
FUSE_HEAD = True                                                          # norm1 -> conv1 (forward)
FUSE_CONV3_DGRAD = True                                                   # conv3's data gradient at the end of the tail backward


def _sub(P, pre):
    """view of the params under a prefix (without the prefix)."""
    n = len(pre)
    return {k[n:]: v for k, v in P.items() if k.startswith(pre)}


def _put(G, pre, g):
    for k, v in g.items():
        G[pre + k] = v


# Weight gradients run on the side stream (kernels.on_side).  A backward helper called on its own joins the side
# branch before it returns (its results are then ordinary current-stream tensors); inside a whole-network backward
# the joins are deferred to the end so the weight-gradient branch overlaps the data-gradient chain of later layers.
_defer_join = 0


class deferred_join:
    def __enter__(self):
        global _defer_join
        _defer_join += 1
        return self

    def __exit__(self, *exc):
        global _defer_join
        _defer_join -= 1
        if _defer_join == 0:
            K.side_join()
        return False


def maybe_join():
    if _defer_join == 0:
        K.side_join()


# Leaf weight gradients of the NAFBlocks (conv1 / conv4 / conv5: nothing downstream in the backward reads them) can be DEFERRED to the
# end of the main backward chain and run on a second HIP stream next to the MASA-encoder backward (kernels.lane): at bs 4 the 1x1 weight
# gradients of the deep levels are L2 / HBM-bound launches of 512 small workgroups, the encoder's 3x3 data / weight gradients are matrix-
# bound launches of 256 - 2048 large ones -- complementary resources, where the NAFBlock chain itself (135 KB of LDS per workgroup) leaves
# no room for a second kernel.  The operands stay referenced until the join (288 GB of HBM: ~14 GB of gradient operands kept alive).
# Only without a gradient exchange: with collectives the buckets are cut in arrival order inside the backward (parallel.GradAllReducer).
DEFER_WGRAD = True
DEFER_LN_FINISH = True                                                  # also the reductions of the LayerNorm-gradient partials
# deferred 1x1 leaf weight gradients of one shape (a level's conv1 / conv4, its conv5) share ONE launch + ONE reduction
# (kernels.wgrad1x1_group, csrc/tdr_wgrad_1x1.hip): no per-launch ramp / prologue / partial write / reduction launch, 1 / 8 of the partials
GROUP_LEAVES = True
# ... and the small finishing reductions queued with them (LayerNorm-gradient partials, depthwise parameter partials, the conv5 / gamma
# parameter gradients behind a grouped weight gradient) run as ONE table-driven launch per kind (kernels.*_multi; the shapes travel in the table): 112 of the 184
# finishing launches of the headline step are the 28 blocks of one level -- beside the MASA-encoder backward they hide, in the data-parallel
# leaf schedule they are serial launches at the level's end
BATCH_FINISH = True
# data-parallel runs: the leaves of a level queued and run (grouped) at the level's end instead of one launch per leaf inside the chain
LEVEL_LEAVES = True
_level_mode = False
FORCE_DP_SCHEDULE = os.environ.get('TDR_FORCE_DP_SCHEDULE', '0') == '1'    # measurement aid: schedule the leaves as a data-parallel run would, on one GPU
SERIAL_LEAVES = False    # measurement aid (bench.py's roofline leg): the deferred leaves on the CURRENT stream, before the main chain
_late = None            # [(prefix, closure -> {name: grad})] while a whole-network backward collects deferred leaves
_late_pre = ''


# RULE for queued leaves: a leaf reads its operands (`keep`, and whatever its closure names) when run_late_leaves() runs it -- after
# the whole main backward chain.  Nothing may write those tensors in place (K.add_ and friends) or rebind the closure's names between
# queueing and the run; every operand must be listed in `keep`.  engine.DEBUG_LEAVES = True checks the tensors' version counters at run time.
DEBUG_LEAVES = False        # (module switch)


def _leaf(keep, fn, G):
    """run a parameter-gradient leaf now (optionally on the side stream), or queue it for the deferred pass"""
    if _late is not None:
        if DEBUG_LEAVES:
            stamp = [(t, t._version) for t in keep if torch.is_tensor(t)]
            inner = fn

            def fn():
                for t, v in stamp:
                    assert t._version == v, 'a queued weight-gradient operand was modified in place before its deferred leaf ran'
                return inner()
        _late.append((_late_pre, fn, keep))
        return
    with K.on_side(*keep):
        for k, v in fn().items():
            G[k] = v                    # (item by item: a GradSink collector acts on __setitem__)


def _leaf_fin(names, fin, G):
    """a leaf that reduces per-workgroup partials: fin() -> the gradients of `names`.  Finishers that carry a `batch` description
    (kernels._ln_partials_finish, kernels.dwsg_bwd) are queued as requests, so that _run_leaves can run all of one shape as one launch"""
    b = getattr(fin, 'batch', None)
    if _late is not None and BATCH_FINISH and b is not None:
        _late.append((_late_pre, ('fin', b[0], b, names, fin), (b[1],)))
        return
    _leaf((), lambda: dict(zip(names, fin())), G)


def _leaf_wgrad1x1(keep, req, post, G, want_db=True):
    """a leaf whose work is ONE 1x1 weight gradient (with the bias gradient unless want_db=False) -- req = (x, dout, Cout, Cin, gate) --
    followed by `post(g, db) -> {name: grad}` (db None without a bias): queued for the grouped launch when leaves are being collected and
    the shape qualifies, an ordinary leaf otherwise"""
    x, dout, Cout, Cin, gate = req
    if _late is not None and GROUP_LEAVES:
        key = K.wgrad1x1_group_key(x, dout, Cout, Cin, gate)
        if key is not None:
            if DEBUG_LEAVES:        # the same in-place check _leaf() installs, for the operands of a grouped request
                stamp = [(t, t._version) for t in keep if torch.is_tensor(t)]
                inner_post = post

                def post(g, db):
                    for t, v in stamp:
                        assert t._version == v, 'a queued weight-gradient operand was modified in place before its grouped leaf ran'
                    return inner_post(g, db)
            _late.append((_late_pre, ('grp', key + (want_db,), req + (want_db,), post), keep))
            return

    def run():
        r = K.conv_wgrad(x, dout, Cout, Cin, 1, gate=gate, want_db=want_db)
        return post(*K.side_keep(*r)) if want_db else post(K.side_keep(r), None)
    _leaf(keep, run, G)


class late_leaves:
    """`with late_leaves(G):` around a whole-network backward: leaves are queued (when allowed) until run_late_leaves().
    With a gradient exchange (G.reducer.collective) nothing is deferred to the end -- the buckets are cut in arrival order inside the
    backward -- but a caller that marks its level boundaries (level_ok=True + level_end(G) after each level) still gets the leaves of
    a LEVEL queued and run together at its end, on the current stream: the 1x1 weight gradients of the level as grouped launches."""

    def __init__(self, G, level_ok=False):
        coll = bool(getattr(getattr(G, 'reducer', None), 'collective', False)) or FORCE_DP_SCHEDULE
        self.level = coll and level_ok and LEVEL_LEAVES
        self.on = DEFER_WGRAD and not K.SIDE_WGRAD and (not coll or self.level)

    def __enter__(self):
        global _late, _grp_seq, _level_mode
        _late = [] if self.on else None
        _level_mode = self.on and self.level
        _grp_seq = 0
        return self

    def __exit__(self, *exc):
        global _late, _level_mode
        _late = None
        _level_mode = False
        return False


def level_end(G):
    """level boundary of a backward pass that exchanges gradients: run what the level queued (grouped), hand its gradients over now"""
    global _late
    if not _level_mode or not _late:
        return
    late, _late = _late, []
    for pre, g in _run_leaves(late, serial=True):
        _put(G, pre, g)
    late.clear()


def set_late_prefix(pre):
    global _late_pre
    _late_pre = pre


_grp_seq = 0            # grouped launches issued so far in this pass (the call-site index of their pinned pointer tables)


def _run_leaves(late, serial=False):
    """the queued leaves on lane 0 (the current stream if serial / SERIAL_LEAVES), 1x1 requests of one shape grouped -> [(prefix, grads)].
    Queue entries: a closure -> {name: grad}; ('grp', key, request, post) -- a 1x1 weight gradient, post(g, db) -> {..}; ('fin', key, batch,
    names, fin) -- a finishing reduction (_leaf_fin)"""
    global _grp_seq
    import contextlib
    ctx = contextlib.nullcontext() if (SERIAL_LEAVES or serial) else K.lane(0, sync=True)
    with ctx:
        groups, fins, outs = {}, {}, {}
        for i, (pre, fn, _) in enumerate(late):
            if isinstance(fn, tuple):
                (fins if fn[0] == 'fin' else groups).setdefault(fn[1], []).append(i)
        scp = []                                                 # posts that are a scaled_conv_param_grads call on a group's result
        for key, idxs in groups.items():                         # one launch + one reduction per shape
            res = K.wgrad1x1_group([late[i][1][2][:5] for i in idxs], seq=_grp_seq, want_db=key[-1])
            _grp_seq += 1
            for i, r in zip(idxs, res):
                if BATCH_FINISH and getattr(late[i][1][3], 'scp', None) is not None:
                    scp.append((i, r))                           # (conv5 / gamma: naf_bwd's post5.scp) -- one launch for all of them below
                else:
                    outs[i] = late[i][1][3](*r)
        if len(scp) > 1:
            items = []
            for i, (g5, s5) in scp:
                w5, b5, gam, c_out, c, _fmt = late[i][1][3].scp
                items.append((g5.view(c_out, c), s5, w5, b5, gam))
            for (i, _r), r3 in zip(scp, K.scaled_conv_param_grads_multi(items, seq=_grp_seq)):
                outs[i] = late[i][1][3].scp[5](*K.side_keep(*r3))
            _grp_seq += 1
        else:
            for i, r in scp:
                outs[i] = late[i][1][3](*r)
        for kind, idxs in fins.items():                          # finishing reductions: one table-driven launch per kind (shapes in the table)
            if len(idxs) == 1:
                i = idxs[0]
                outs[i] = dict(zip(late[i][1][3], late[i][1][4]()))
                continue
            multi = K.pair_sum_partials_multi if kind == 'ln' else K.dw_param_finish_multi
            for i, r in zip(idxs, multi([late[i][1][2][1:] for i in idxs], seq=_grp_seq)):
                outs[i] = dict(zip(late[i][1][3], r))
            _grp_seq += 1
        return [(pre, outs[i] if isinstance(fn, tuple) else fn()) for i, (pre, fn, _) in enumerate(late)]


def run_late_leaves(G, main_chain):
    """deferred leaves on lane 0, `main_chain()` on the current stream, join, then hand the gradients to the collector in order"""
    global _late, _grp_seq
    if _level_mode:                 # gradient exchange: the remainder of the last level, then the main chain -- nothing runs beside it
        level_end(G)
        _late = None
        _grp_seq = 0
        main_chain()
        return
    late, _late = _late, None
    if not late:
        _grp_seq = 0
        main_chain()
        return
    results = _run_leaves(late)
    _grp_seq = 0
    main_chain()
    K.lanes_join()
    for pre, g in results:
        _put(G, pre, g)
    late.clear()      # (operands referenced until here: the allocator cannot recycle them under a running lane kernel)


# ---------------------------------------------------------------------------
# NAFBlock / NAFResFuseBlock   models/archs/network_nafnet_guided_arch.py:178-302
# ---------------------------------------------------------------------------
def naf_fwd(x, P, c_out=None):
    """x [N,c,H,W] -> [N,c_out,H,W] (c_out<c: only the first c_out output channels are
    produced, the `[:, :chan]` slice of :719/:727 folded into conv5)."""
    N, c, H, W = x.shape
    c_out = c if c_out is None else c_out
    wp, mp, *_ = K.pack_weights(P['conv1.weight'], PACK_FWD)
    if FUSE_HEAD and K.naf_tail_supported(c, H * W) and wp.fmt in (K.FMT_HX2, K.FMT_BX3):
        # norm1 -> conv1 in one launch (the workgroup that owns 64 pixels x all channels reduces the statistics itself)
        xn, mu1, rs1, t1 = K.naf_head_fwd(x, P['norm1.weight'], P['norm1.bias'], LN_EPS, wp, P['conv1.bias'])
    else:
        xn, mu1, rs1 = K.layernorm2d_fwd(x, P['norm1.weight'], P['norm1.bias'], LN_EPS)
        t1 = K.conv_forward(xn, wp, mp, 2 * c, 1, bias=P['conv1.bias'])
    g, pooled = K.dwsg_fwd(t1, P['conv2.weight'], P['conv2.bias'])
    s = K.sca_fwd(pooled, P['sca.1.weight'], P['sca.1.bias'])
    if FUSE_TAIL and K.naf_tail_supported(c, H * W, c_out) and x.is_contiguous():
        # conv3 -> norm2 -> conv4 -> SimpleGate -> conv5 in one launch (one workgroup per 64 pixels x all channels)
        w3p, w4p = (K.pack_weights(P[k], PACK_FWD)[0] for k in ('conv3.weight', 'conv4.weight'))
        w5p = K.pack_weights(P['conv5.weight'][:c_out], PACK_FWD)[0]
        out, y, mu2, rs2, yn, t4 = K.naf_tail_fwd(g, s, x, w3p, P['conv3.bias'], P['beta'].view(-1), P['norm2.weight'],
                                                  P['norm2.bias'], LN_EPS, w4p, P['conv4.bias'], w5p, P['conv5.bias'],
                                                  P['gamma'].view(-1), c_out=c_out)
        return out, (x, xn, mu1, rs1, t1, g, pooled, s, y, yn, mu2, rs2, t4, c_out)
    wp, mp, *_ = K.pack_weights(P['conv3.weight'], PACK_FWD)
    y = K.conv_forward(g, wp, mp, c, 1, kscale=s, bias=P['conv3.bias'], scale=P['beta'].view(-1), res=x)
    yn, mu2, rs2 = K.layernorm2d_fwd(y, P['norm2.weight'], P['norm2.bias'], LN_EPS)
    wp, mp, *_ = K.pack_weights(P['conv4.weight'], PACK_FWD)
    t4 = K.conv_forward(yn, wp, mp, 2 * c, 1, bias=P['conv4.bias'])
    wp, mp, *_ = K.pack_weights(P['conv5.weight'][:c_out], PACK_FWD)
    out = K.conv_forward(t4, wp, mp, c_out, 1, gate=True, bias=P['conv5.bias'], scale=P['gamma'].view(-1), res=y)
    saved = (x, xn, mu1, rs1, t1, g, pooled, s, y, yn, mu2, rs2, t4, c_out)
    return out, saved


def _dgrad_is_hx2():
    """data-gradient weights are packed in the fp16-split layout only inside a loss-scaled backward pass (kernels.GRAD_SCALED)"""
    return K.MATH == 'hx2' and K.GRAD_SCALED


def _dgrad_fused_ok():
    """the fused backward chains take the fp16-pair packs of a loss-scaled hx2 backward or the bf16-triple packs of TDR_MATH=bx3
    (unscaled gradients, fp32 range); an hx2 backward outside a scaled step and the other modes run the per-op launches"""
    return _dgrad_is_hx2() or K.MATH == 'bx3'


def naf_bwd(dout, P, saved):
    x, xn, mu1, rs1, t1, g, pooled, s, y, yn, mu2, rs2, t4, c_out = saved
    N, c, H, W = x.shape
    dev = x.device
    G = {}
    beta, gamma = P['beta'].view(-1), P['gamma'].view(-1)
    late = _late is not None and DEFER_LN_FINISH
    # ---- conv5 / gamma chain (parameter gradients only: a leaf off the data-gradient chain)
    def fmt5(dw5, db5, dgam):
        g = {}
        if c_out == c:
            g['conv5.weight'], g['conv5.bias'], g['gamma'] = dw5.view(c, c, 1, 1), db5, dgam.view(1, c, 1, 1)
        else:
            fw = torch.zeros(c, c, 1, 1, dtype=torch.float32, device=dev)
            fb = torch.zeros(c, dtype=torch.float32, device=dev)
            fg = torch.zeros(1, c, 1, 1, dtype=torch.float32, device=dev)
            K.copy_rows(dw5, 0, fw, 0, 1, c_out * c)
            K.copy_rows(db5, 0, fb, 0, 1, c_out)
            K.copy_rows(dgam, 0, fg, 0, 1, c_out)
            g['conv5.weight'], g['conv5.bias'], g['gamma'] = fw, fb, fg
        return g

    def post5(G5, S5):
        return fmt5(*K.side_keep(*K.scaled_conv_param_grads(G5.view(c_out, c), S5, P['conv5.weight'], P['conv5.bias'], gamma)))
    post5.scp = (P['conv5.weight'], P['conv5.bias'], gamma, c_out, c, fmt5)      # (what _run_leaves needs to batch it with its level's others)
    _leaf_wgrad1x1((t4, dout), (t4, dout, c_out, c, True), post5, G)
    fused = FUSE_TAIL and K.naf_tail_supported(c, H * W, c_out) and dout.is_contiguous() and \
        _dgrad_fused_ok()
    if fused:
        # conv5 dgrad -> SimpleGate bwd -> conv4 dgrad -> norm2 bwd (+ skip) in one launch
        w5t, w4t = K.pack_weights(P['conv5.weight'][:c_out], PACK_DGRAD_S1)[0], K.pack_weights(P['conv4.weight'], PACK_DGRAD_S1)[0]
        if FUSE_CONV3_DGRAD:
            w3t = K.pack_weights(P['conv3.weight'], PACK_DGRAD_S1)[0]
            dy, dt4, gw2, gb2, dgp = K.naf_tail_bwd(dout, gamma, t4, y, mu2, rs2, P['norm2.weight'], w5t, w4t, w3tp=w3t, beta=beta,
                                                    sca=s.contiguous(), defer_finish=late)
        else:
            dy, dt4, gw2, gb2 = K.naf_tail_bwd(dout, gamma, t4, y, mu2, rs2, P['norm2.weight'], w5t, w4t, defer_finish=late)
        if late:        # the reduction of the per-workgroup LayerNorm-gradient partials is a leaf too (gw2: closure over its private buffer)
            _leaf_fin(('norm2.weight', 'norm2.bias'), gw2, G)
        else:
            G['norm2.weight'], G['norm2.bias'] = gw2, gb2
    else:
        wp, mp, *_ = K.pack_weights(P['conv5.weight'][:c_out], PACK_DGRAD_S1)
        dt4 = K.conv_forward(dout, wp, mp, c, 1, epi=EPI_GATEBWD, kscale=gamma, aux=t4)
    # ---- conv4
    _leaf_wgrad1x1((yn, dt4), (yn, dt4, 2 * c, c, False),
                   lambda g4, b4: {'conv4.weight': g4.view(2 * c, c, 1, 1), 'conv4.bias': b4}, G)
    if not fused:
        wp, mp, *_ = K.pack_weights(P['conv4.weight'], PACK_DGRAD_S1)
        dyn = K.conv_forward(dt4, wp, mp, c, 1)
        # ---- norm2 (+ residual branch of `y + x*gamma`)
        dy, G['norm2.weight'], G['norm2.bias'] = K.layernorm2d_bwd(dyn, y, mu2, rs2, P['norm2.weight'], add=dout)
    # ---- conv3 / SCA / beta chain
    G3, S3 = K.conv_wgrad(g, dy, c, c, 1, per_image=True, want_db=True)
    dw3, db3, dbeta, dwsca, dbsca, dpooled = K.sca_bwd(G3, S3, P['conv3.weight'], P['conv3.bias'], beta, s, pooled,
                                                       P['sca.1.weight'])
    G['conv3.weight'], G['conv3.bias'], G['beta'] = dw3, db3, dbeta
    G['sca.1.weight'], G['sca.1.bias'] = dwsca, dbsca
    # ---- depthwise + SimpleGate
    if fused and FUSE_CONV3_DGRAD:
        # conv3's data gradient already left the tail kernel; its pooled-gradient term joins as a per-plane constant
        dt1, gdw, gdb = K.dwsg_bwd(dgp, t1, P['conv2.weight'], P['conv2.bias'], dg_bias=dpooled, dg_bias_mul=1.0 / (H * W),
                                   defer_finish=late)
    else:
        wp, mp, *_ = K.pack_weights(P['conv3.weight'], PACK_DGRAD_S1)
        dg = K.conv_forward(dy, wp, mp, c, 1, kscale=beta, scale=s, bias2=dpooled, bias2_mul=1.0 / (H * W))
        dt1, gdw, gdb = K.dwsg_bwd(dg, t1, P['conv2.weight'], P['conv2.bias'], defer_finish=late)
    if callable(gdw):       # the finish of the depthwise parameter gradients: one more leaf
        _leaf_fin(('conv2.weight', 'conv2.bias'), gdw, G)
    else:
        G['conv2.weight'], G['conv2.bias'] = gdw, gdb
    # ---- conv1
    _leaf_wgrad1x1((xn, dt1), (xn, dt1, 2 * c, c, False),
                   lambda g1, b1: {'conv1.weight': g1.view(2 * c, c, 1, 1), 'conv1.bias': b1}, G)
    if FUSE_TAIL and K.naf_tail_supported(c, H * W) and _dgrad_fused_ok() and x.is_contiguous() and dy.is_contiguous():
        # conv1 dgrad -> norm1 bwd (+ dy) in one launch
        w1t = K.pack_weights(P['conv1.weight'], PACK_DGRAD_S1)[0]
        dx, gw1, gb1 = K.naf_head_bwd(dt1, x, mu1, rs1, P['norm1.weight'], w1t, dy, defer_finish=late)
        if late:
            _leaf_fin(('norm1.weight', 'norm1.bias'), gw1, G)
        else:
            G['norm1.weight'], G['norm1.bias'] = gw1, gb1
    else:
        wp, mp, *_ = K.pack_weights(P['conv1.weight'], PACK_DGRAD_S1)
        dxn = K.conv_forward(dt1, wp, mp, c, 1)
        dx, G['norm1.weight'], G['norm1.bias'] = K.layernorm2d_bwd(dxn, x, mu1, rs1, P['norm1.weight'], add=dy)
    maybe_join()
    return dx, G


def naf_seq_fwd(x, P, pre, n, c_out_last=None, local=None):
    """local = (k1, k2): TLSC inference (naf_fwd_local), nothing saved"""
    saved = []
    for i in range(n):
        if local is not None:
            x = naf_fwd_local(x, _sub(P, f'{pre}{i}.'), *local)
            continue
        x, sv = naf_fwd(x, _sub(P, f'{pre}{i}.'), c_out_last if i == n - 1 else None)
        saved.append(sv)
    return x, saved


def naf_fwd_local(x, P, k1, k2):
    """NAFBlock forward with the SCA branch's global average pool replaced by TLSC's local box mean of k1 x k2 pixels
    (models/archs/nafnet_local_arch.py:10-75, `replace_layers`): the pooled statistic -- and with it the channel attention --
    becomes a per-pixel map, so `x * sca(x)` (:192) is an element-wise product of two maps, folded into conv3's operand load as
    the gate product of the concatenation [g ; sca(pool(g))].  Inference only (the reference wraps the network in eval /
    no_grad, network_nafnet_guided_arch.py:756-768).  Where the box covers the whole map the reference falls back to
    F.adaptive_avg_pool2d(x, 1) (:43-44): that is the ordinary block."""
    N, c, H, W = x.shape
    if k1 >= H and k2 >= W:
        return naf_fwd(x, P)[0]
    wp, mp, *_ = K.pack_weights(P['conv1.weight'], PACK_FWD)
    xn, _, _ = K.layernorm2d_fwd(x, P['norm1.weight'], P['norm1.bias'], LN_EPS)
    t1 = K.conv_forward(xn, wp, mp, 2 * c, 1, bias=P['conv1.bias'])
    g, _ = K.dwsg_fwd(t1, P['conv2.weight'], P['conv2.bias'])
    cat = torch.empty(N, 2 * c, H, W, dtype=torch.float32, device=x.device)
    K.copy_rows(g, c * H * W, cat, 2 * c * H * W, N, c * H * W)
    wp, mp, *_ = K.pack_weights(P['sca.1.weight'], PACK_FWD)
    K.conv_forward(K.local_avgpool(g, k1, k2), wp, mp, c, 1, bias=P['sca.1.bias'], out=cat[:, c:])
    wp, mp, *_ = K.pack_weights(P['conv3.weight'], PACK_FWD)
    y = K.conv_forward(cat, wp, mp, c, 1, gate=True, bias=P['conv3.bias'], scale=P['beta'].view(-1), res=x)
    yn, _, _ = K.layernorm2d_fwd(y, P['norm2.weight'], P['norm2.bias'], LN_EPS)
    wp, mp, *_ = K.pack_weights(P['conv4.weight'], PACK_FWD)
    t4 = K.conv_forward(yn, wp, mp, 2 * c, 1, bias=P['conv4.bias'])
    wp, mp, *_ = K.pack_weights(P['conv5.weight'], PACK_FWD)
    return K.conv_forward(t4, wp, mp, c, 1, gate=True, bias=P['conv5.bias'], scale=P['gamma'].view(-1), res=y)


def naf_seq_bwd(dout, P, pre, n, saved, G):
    global _late_pre
    for i in reversed(range(n)):
        _late_pre = f'{pre}{i}.'
        dout, g = naf_bwd(dout, _sub(P, f'{pre}{i}.'), saved[i])
        _put(G, f'{pre}{i}.', g)
    _late_pre = ''
    return dout


# ---------------------------------------------------------------------------
# dense convs: intro / ending / downs / ups (:429-434, :449-451, :468-473)
# ---------------------------------------------------------------------------
def conv_fwd(x, w, b, stride, pad, res=None, relu=False, out=None):
    Cout, Cin, KH, _ = w.shape
    wp, mp, *_ = K.pack_weights(w, PACK_FWD)
    out = K.conv_forward(x, wp, mp, Cout, KH, stride=stride, pad=pad, bias=b, res=res, relu=relu, out=out)
    return out


def conv_bwd(dout, x, w, stride, pad, need_dx=True, add_to_dx=None, bias=True, into=None):
    """returns (dx or None, dw, db); db is None for a bias-free conv (bias=False).
    into = (G, weight name, bias name or None): the parameter gradients go to the collector instead -- as a leaf (_leaf: deferred to
    the second stream when a whole-network backward collects leaves) -- and (dx, None, None) is returned."""
    Cout, Cin, KH, _ = w.shape

    def leaf():
        if bias:
            gw, gb = K.conv_wgrad(x, dout, Cout, Cin, KH, stride=stride, pad=pad, want_db=True)
        else:
            gw, gb = K.conv_wgrad(x, dout, Cout, Cin, KH, stride=stride, pad=pad), None
        return gw.view(Cout, Cin, KH, KH), gb
    dw = db = None
    if into is not None:
        Gc, wname, bname = into
        set_late_prefix('')

        def leaf_named():
            gw, gb = leaf()
            return {wname: gw} if (gb is None or bname is None) else {wname: gw, bname: gb}
        _leaf((x, dout), leaf_named, Gc)
    else:
        with K.on_side(x, dout):
            dw, db = leaf()
    dx = None
    if need_dx:
        N, _, OH, OW = dout.shape
        if stride == 1:
            wp, mp, *_ = K.pack_weights(w, PACK_DGRAD_S1)
            dx = K.conv_forward(dout, wp, mp, Cin, KH, pad=KH - 1 - pad, res=add_to_dx)
        elif KH == 2 and stride == 2 and pad == 0:
            wp, mp, *_ = K.pack_weights(w, PACK_DGRAD_2X2S2)
            dx = K.conv_forward(dout, wp, mp, 4 * Cin, 1, epi=EPI_PSHUF, res=add_to_dx)
        elif KH == 3 and stride == 2 and pad == 1:
            wp, mp, *_ = K.pack_weights(w, PACK_DGRAD_3X3S2)
            dx = K.conv_forward(dout, wp, mp, 4 * Cin, 2, pad=0, OH=OH, OW=OW, epi=EPI_PSHUF, res=add_to_dx)
        else:
            raise NotImplementedError(f'conv dgrad KH={KH} stride={stride} pad={pad}')
    maybe_join()
    return dx, dw, db


def up_fwd(x, w, skip):
    """ups: 1x1 (C->2C, no bias) + PixelShuffle(2), then `+ enc_skip` (:733-734)."""
    C2 = w.shape[0]
    wp, mp, *_ = K.pack_weights(w, PACK_FWD)
    return K.conv_forward(x, wp, mp, C2, 1, epi=EPI_PSHUF, res=skip)


def up_bwd(dout, x, w, into=None):
    """into = (G, weight name): as in conv_bwd"""
    C2, Cc = w.shape[0], w.shape[1]
    dT = K.pixel_unshuffle2(dout)
    dw = None
    if into is not None:
        set_late_prefix('')
        _leaf((x, dT), lambda: {into[1]: K.conv_wgrad(x, dT, C2, Cc, 1).view(C2, Cc, 1, 1)}, into[0])
    else:
        with K.on_side(x, dT):
            dw = K.conv_wgrad(x, dT, C2, Cc, 1).view(C2, Cc, 1, 1)
    wp, mp, *_ = K.pack_weights(w, PACK_DGRAD_S1)
    dx = K.conv_forward(dT, wp, mp, Cc, 1)
    maybe_join()
    return dx, dw


# ---------------------------------------------------------------------------
# MASA feature encoder (Encoder + ResidualBlock, :44-59, :110-143)
# ---------------------------------------------------------------------------
def _enc_counts(ext):
    return [ext[0], ext[1], ext[2], ext[2], ext[2]]


# The ResidualBlock convolutions run on PRE-SPLIT activations (kernels.P16, csrc/tdr_conv_p16.hip / tdr_wgrad_p16.hip) when the
# step's arithmetic is the 2-way fp16 split in both passes (TDR_MATH=hx2 inside a loss-scaled step) and the level's channel
# count is a multiple of 16: conv1 / conv2 read and write the fp16 pair planes (the same 4 bytes per element as the fp32
# tensors they replace: `h`, the block inputs and, in the backward pass, `dh` and the gradient stream exist ONLY as pairs),
# both weight gradients read them through transposed LDS reads, and the ReLU masks are the sign of the head plane.  A level
# enters the format through one conversion of conv_L's output (forward) and of the incoming feature gradient (backward) and
# leaves it as fp32 (feats[lvl] for the MASA kernels / the next conv_L, the gradient for conv_L's backward).
# (P16_ON = False keeps the fp32 tensors + per-consumer split of rounds 1-3.)
P16_ON = True
# narrower levels (C = 32: one 32-row m-tile, 18 (group, tap) steps) keep the fp32 kernels until the weights-stationary variant exists
# (bf16 triple planes, TDR_MATH=bx3: the C = 32 level too -- its convolution is a wash at 6 bytes per element (292 vs 283 us per launch), its
# weight gradient is not (232 vs 273 us): -0.5 ms per step, same-box A/B profiles/r5/sweep_a.log)
P16_MIN_C = None


def _p16_level(Cc, n_blocks):
    """plane tensors at this level: fp16 pairs inside a loss-scaled hx2 step, bf16 triples under TDR_MATH=bx3 (kernels.plane_fmt)"""
    fmt = K.plane_fmt()
    min_c = P16_MIN_C if P16_MIN_C is not None else (32 if fmt == K.FMT_BX3 else 64)
    return P16_ON and n_blocks > 0 and fmt is not None and K.p16_supported(Cc) and Cc >= min_c


def encoder_fwd(x, P, pre, ext_n_blocks, levels=5):
    """returns ([f1..f_levels], saved).  levels=4: the Restormer-ref file's own 4-level Encoder
    (network_restormer_guided_arch.py:99-133)."""
    feats, saved = [], []
    cnt = _enc_counts(ext_n_blocks)
    for lvl in range(levels):
        k = lvl + 1
        xin = x
        a = conv_fwd(xin, P[f'{pre}conv_L{k}.weight'], P[f'{pre}conv_L{k}.bias'], 1 if lvl == 0 else 2, 1, relu=True)
        blocks = []
        x = a
        Cc = a.shape[1]
        if _p16_level(Cc, cnt[lvl]):
            # The FORWARD residual stream stays fp32 (conv2 writes the fp32 sum next to its pair image): the pair of x is x rounded
            # to ~23 bits, and a 1e-7 perturbation of the stream flips a handful of ReLU decisions h > 0 per tensor -- each flip moves
            # a conv1 gradient element by a whole term (measured 1.8e-4 of the tensor maximum against the fp32-tensor path,
            # profiles/r4/diag_p16_grads.log).  With the fp32 stream the forward pass is bit-identical to the fp32-tensor kernels.
            # bf16 TRIPLE planes (TDR_MATH=bx3) hold every fp32 value exactly (h + m + l == x): there the residual stream itself
            # lives in the planes and only the level's output is also written as fp32 -- bit-identical to the fp32-tensor kernels.
            fmt = K.plane_fmt()
            tri = fmt == K.FMT_BX3
            x16, x32 = K.p16_from_f32(a, fmt=fmt), a
            for i in range(cnt[lvl]):
                bp = f'{pre}blk_L{k}.{i}.'
                wp1, mp1, *_ = K.pack_weights(P[bp + 'conv1.weight'], PACK_FWD)
                wp2, mp2, *_ = K.pack_weights(P[bp + 'conv2.weight'], PACK_FWD)
                last = i == cnt[lvl] - 1
                _, h16 = K.conv3x3_p16(x16, wp1, mp1, Cc, bias=P[bp + 'conv1.bias'], relu=True, want32=False, want16=True)
                o32, o16 = K.conv3x3_p16(h16, wp2, mp2, Cc, bias=P[bp + 'conv2.bias'], res=x16 if tri else x32,
                                         want32=last or not tri, want16=not last)
                blocks.append((x16, h16))
                x16, x32 = o16, o32
            x = x32
        else:
            for i in range(cnt[lvl]):
                bp = f'{pre}blk_L{k}.{i}.'
                h = conv_fwd(x, P[bp + 'conv1.weight'], P[bp + 'conv1.bias'], 1, 1, relu=True)
                o = conv_fwd(h, P[bp + 'conv2.weight'], P[bp + 'conv2.bias'], 1, 1, res=x)
                blocks.append((x, h))
                x = o
        feats.append(x)
        saved.append((xin, a, blocks))
    return feats, saved


def encoder_bwd(dfeats, P, pre, ext_n_blocks, saved, G):
    """dfeats: list of per-level grads (or None).  Input-image gradient is not needed."""
    cnt = _enc_counts(ext_n_blocks)
    dnext = None                      # gradient flowing from level lvl+1 into feats[lvl]
    with deferred_join():
        _encoder_bwd_levels(dfeats, P, pre, cnt, saved, G, dnext)
    maybe_join()
    return None


def _encoder_bwd_levels(dfeats, P, pre, cnt, saved, G, dnext):
    for lvl in reversed(range(len(dfeats))):
        k = lvl + 1
        xin, a, blocks = saved[lvl]
        d = dnext if dnext is not None else dfeats[lvl]      # dnext already contains dfeats[lvl] (add_to_dx below)
        if d is None:
            continue
        if blocks and isinstance(blocks[0][0], K.P16):
            d16, d32 = K.p16_from_f32(d, fmt=blocks[0][0].fmt), d
            for i in reversed(range(cnt[lvl])):
                bp = f'{pre}blk_L{k}.{i}.'
                x16, h16 = blocks[i]
                w1, w2 = P[bp + 'conv1.weight'], P[bp + 'conv2.weight']
                Cc = w1.shape[0]
                with K.on_side(h16.buf, d16.buf):
                    gw, G[bp + 'conv2.bias'] = K.wgrad3x3_p16(h16, d16, want_db=True)
                    G[bp + 'conv2.weight'] = gw.view(Cc, Cc, 3, 3)
                wp, mp, *_ = K.pack_weights(w2, PACK_DGRAD_S1)
                _, dh16 = K.conv3x3_p16(d16, wp, mp, Cc, mask=h16, want32=False, want16=True)
                with K.on_side(x16.buf, dh16.buf):
                    gw, G[bp + 'conv1.bias'] = K.wgrad3x3_p16(x16, dh16, want_db=True)
                    G[bp + 'conv1.weight'] = gw.view(Cc, Cc, 3, 3)
                wp, mp, *_ = K.pack_weights(w1, PACK_DGRAD_S1)
                # the first block's input is the level's ReLU output `a`: its mask rides on this epilogue (conv + res, then mask);
                # the gradient leaves the level as fp32 (conv_L's backward), stays a pair otherwise
                first = i == 0
                d32, d16 = K.conv3x3_p16(dh16, wp, mp, Cc, res=d32 if d32 is not None else d16, mask=a if first else None,
                                         want32=first, want16=not first)
            d = d32
        else:
            for i in reversed(range(cnt[lvl])):
                bp = f'{pre}blk_L{k}.{i}.'
                x_in, h = blocks[i]
                w1, w2 = P[bp + 'conv1.weight'], P[bp + 'conv2.weight']
                Cc = w1.shape[0]
                with K.on_side(h, d):
                    gw, G[bp + 'conv2.bias'] = K.conv_wgrad(h, d, Cc, Cc, 3, pad=1, want_db=True)
                    G[bp + 'conv2.weight'] = gw.view(Cc, Cc, 3, 3)
                wp, mp, *_ = K.pack_weights(w2, PACK_DGRAD_S1)
                dh = K.conv_forward(d, wp, mp, Cc, 3, pad=1, mask=h)
                with K.on_side(x_in, dh):
                    gw, G[bp + 'conv1.bias'] = K.conv_wgrad(x_in, dh, Cc, Cc, 3, pad=1, want_db=True)
                    G[bp + 'conv1.weight'] = gw.view(Cc, Cc, 3, 3)
                wp, mp, *_ = K.pack_weights(w1, PACK_DGRAD_S1)
                # the first block's input is the level's ReLU output `a`: its mask rides on this epilogue (conv + res, then mask)
                d = K.conv_forward(dh, wp, mp, Cc, 3, pad=1, res=d, mask=a if i == 0 else None)
        dpre = d if cnt[lvl] > 0 else K.relu_bwd(d, a)
        w = P[f'{pre}conv_L{k}.weight']
        # the feature gradient of the level below joins in the data-gradient epilogue instead of a separate add
        dnext, G[f'{pre}conv_L{k}.weight'], G[f'{pre}conv_L{k}.bias'] = conv_bwd(
            dpre, xin, w, 1 if lvl == 0 else 2, 1, need_dx=(lvl > 0), add_to_dx=dfeats[lvl - 1] if lvl > 0 else None)
    maybe_join()
    return None


# ---------------------------------------------------------------------------
# MASA match + transfer (:597-707)
# ---------------------------------------------------------------------------
class MasaGeom:
    def __init__(self, h, w, hr, wr, n_enc, lr_block_size, ref_down_block_size, dilations):
        padder = 2 ** n_enc           # padder_size (:419; Restormer-ref: 2**3, network_restormer_guided_arch.py:546)
        self.px = w // padder // lr_block_size
        self.py = h // padder // lr_block_size
        self.kx = w // padder // self.px
        self.ky = h // padder // self.py
        self.dia_x = 2 * int(wr // padder // (2 * self.px) * ref_down_block_size) + 1
        self.dia_y = 2 * int(hr // padder // (2 * self.py) * ref_down_block_size) + 1
        if self.dia_x != self.dia_y or self.kx != self.ky:
            raise ValueError('MASA geometry must be square (the reference only runs for square geometry, :668-669)')
        self.dilations = list(dilations)
        self.P = self.py * self.px
        self.K = self.kx
        self.side = self.dia_x + 2


class Pyramids:
    """The two MASA feature pyramids of a forward pass (`feat_lq = masa_enc(inp)`, `feat_ref = masa_enc(ref)`, :617-618).
    When lq and ref pad to the same size (training: the DINO window match makes them equal) both images are stacked
    into ONE 2N batch and the encoder runs once; otherwise (validation / inference: the full generated reference
    against an lq of any size, image_restoration_ref_model.py:286-330) it runs once per tensor.  `lq_deep` /
    `ref_feats` are what the match-and-transfer stage reads in both cases."""
    __slots__ = ('N', 'stacked', 'inp_p', 'geo', 'feats', 'sv_enc', 'lq_deep', 'ref_feats', 'levels')


def pyramids_fwd(P, cfg, inp, ref, padder_log2, levels):
    N, Ci, H0, W0 = inp.shape
    mult = (2 ** padder_log2) * cfg['lr_block_size']
    Hp, Wp = -(-H0 // mult) * mult, -(-W0 // mult) * mult
    Hr0, Wr0 = ref.shape[-2:]
    Hrp, Wrp = -(-Hr0 // mult) * mult, -(-Wr0 // mult) * mult
    if ref.shape[0] != N:
        raise ValueError('inp and ref must have the same batch size')
    py = Pyramids()
    py.N, py.levels = N, levels
    py.geo = MasaGeom(Hp, Wp, Hrp, Wrp, padder_log2, cfg['lr_block_size'], cfg['ref_down_block_size'], cfg['dilations'])
    py.stacked = (Hrp, Wrp) == (Hp, Wp)
    if py.stacked:
        both = torch.empty(2 * N, Ci, Hp, Wp, dtype=torch.float32, device=inp.device)
        # zero-pad (:576-585) and stack [lq; ref] so masa_enc runs once over 2N images
        _pad_into(inp.contiguous(), both[:N])
        _pad_into(ref.contiguous(), both[N:])
        py.inp_p = both[:N]
        py.feats, py.sv_enc = encoder_fwd(both, P, 'masa_enc.', cfg['ext_n_blocks'], levels=levels)
        py.lq_deep = py.feats[levels - 1][:N]
        py.ref_feats = [f[N:] for f in py.feats]
    else:
        py.inp_p = torch.empty(N, Ci, Hp, Wp, dtype=torch.float32, device=inp.device)
        ref_p = torch.empty(N, Ci, Hrp, Wrp, dtype=torch.float32, device=inp.device)
        _pad_into(inp.contiguous(), py.inp_p)
        _pad_into(ref.contiguous(), ref_p)
        fl, svl = encoder_fwd(py.inp_p, P, 'masa_enc.', cfg['ext_n_blocks'], levels=levels)
        fr, svr = encoder_fwd(ref_p, P, 'masa_enc.', cfg['ext_n_blocks'], levels=levels)
        py.feats, py.sv_enc = (fl, fr), (svl, svr)
        py.lq_deep, py.ref_feats = fl[levels - 1], fr
    return py, (H0, W0, Hp, Wp)


def pyramids_bwd(dwarp, py, P, cfg, sv_masa, G):
    """MASA match/transfer backward + masa_enc backward (both pyramids share the encoder weights: one pass over the
    stacked batch, or two passes whose weight gradients are added)."""
    N, L = py.N, py.levels
    if py.stacked:
        dfeats = [torch.zeros_like(f) for f in py.feats]
        masa_bwd(dwarp, py.lq_deep, py.ref_feats, N, py.geo, sv_masa, dfeats[L - 1][:N], [d[N:] for d in dfeats])
        encoder_bwd(dfeats, P, 'masa_enc.', cfg['ext_n_blocks'], py.sv_enc, G)
        return
    fl, fr = py.feats
    dlq = torch.empty_like(fl[L - 1])
    dref = [torch.zeros_like(f) for f in fr]
    masa_bwd(dwarp, py.lq_deep, py.ref_feats, N, py.geo, sv_masa, dlq, dref)
    Gl, Gr = {}, {}
    encoder_bwd([None] * (L - 1) + [dlq], P, 'masa_enc.', cfg['ext_n_blocks'], py.sv_enc[0], Gl)
    encoder_bwd(dref, P, 'masa_enc.', cfg['ext_n_blocks'], py.sv_enc[1], Gr)
    for k, g in Gr.items():          # levels above the deepest only see the ref pyramid (lq feeds the search alone)
        G[k] = K.add_(g.contiguous().view(1, -1), Gl[k].contiguous().view(1, -1)).view(g.shape) if k in Gl else g


def masa_fwd(lq4, ref_feats, N, geo, outs=None):
    """lq4: deepest lq feature map [N,C,H,W]; ref_feats: the ref pyramid (finest first; 5 levels for NAFNet-ref, 4 for
    Restormer-ref), each [N,C_l,Hr_l,Wr_l].  Returns (warp list finest->coarsest like the reference's warp_ref_l,
    saved).  outs: optional per-level destination views (the second half of the fusion blocks' concat buffers)."""
    P, Kk, side = geo.P, geo.K, geo.side
    L = len(ref_feats)
    ref4 = ref_feats[L - 1]
    _, Cc, H, W = lq4.shape
    Hr, Wr = ref4.shape[-2:]
    dev = lq4.device
    lrb = K.lr_blocks_fwd(lq4, geo.py, geo.px, Kk, Kk)                    # [N*P, C, K+2, K+2]
    # ---- coarse search (:515-536) on the MFMA conv with the LR centre taps as filters
    ND = len(geo.dilations)
    R = Hr * Wr
    dots = torch.empty(ND, N, P, Hr, Wr, dtype=torch.float32, device=dev)
    invq = torch.empty(ND, N * P, dtype=torch.float32, device=dev)
    invk = torch.empty(ND, N, R, dtype=torch.float32, device=dev)
    cc = (Kk + 2) // 2
    for di, d in enumerate(geo.dilations):
        wp, mp, per_b = K.pack_patches(lrb, P, 1, 1, 1, d, cc - d)
        K.conv_forward(ref4, wp, mp, P, 3, dil=d, pad=d, wp_ns=per_b, out=dots[di])
        K.patch_inv_norm(lrb, 1, 1, dil=d, off=cc - d, out=invq[di])
        K.patch_inv_norm(ref4, Hr, Wr, dil=d, pad=d, out=invk[di])
    index, y1, x1 = K.coarse_argmax_box(dots, invq, invk, N, P, Hr, Wr, geo.dia_x)
    # ---- fine search (:495-513) inside the matched ref block
    refb = K.gather_ref_block(ref4, y1, x1, P, side, 1)                   # [N*P, C, side, side]
    wp, mp, per_b = K.pack_patches(lrb, 1, Kk, Kk, 1, 1, 0)
    R1 = side - 2
    fdots = K.conv_forward(refb, wp, mp, Kk * Kk, 3, pad=0, wp_ns=per_b)  # [N*P, K*K, R1, R1]
    finvq = K.patch_inv_norm(lrb, Kk, Kk)                                 # [N*P, K, K]
    finvk = K.patch_inv_norm(refb, R1, R1)
    index_all, soft_att = K.fine_argmax(fdots, finvq, finvk, N * P, Kk * Kk, R1 * R1)
    # ---- transfer at every scale, reading the ref features directly
    warp = []
    for lvl in range(L):
        s = 2 ** (L - 1 - lvl)
        warp.append(K.transfer_fwd(ref_feats[lvl], y1, x1, index_all, soft_att, geo.py, geo.px, Kk, side, s,
                                   out=None if outs is None else outs[lvl]))
    saved = (lrb, refb, finvq, finvk, index, y1, x1, index_all, soft_att)
    return warp, saved


def masa_bwd(dwarp, lq4, ref_feats, N, geo, saved, dlq_out, dref_out):
    """dlq_out [N,C,H,W] receives the gradient w.r.t. the deepest lq features (overwritten); dref_out: per-level
    ZERO-INITIALISED tensors (dense batch slices are fine) that receive the gradients w.r.t. the ref pyramid."""
    lrb, refb, finvq, finvk, index, y1, x1, index_all, soft_att = saved
    P, Kk, side = geo.P, geo.K, geo.side
    L = len(ref_feats)
    dev = lq4.device
    datt = torch.zeros(N * P, Kk * Kk, dtype=torch.float32, device=dev)
    for lvl in range(L):
        s = 2 ** (L - 1 - lvl)
        K.transfer_bwd(dwarp[lvl], ref_feats[lvl], y1, x1, index_all, soft_att, geo.py, geo.px, Kk, side, s,
                       dref_out[lvl], datt)
    dlrb, drefb = K.fine_search_bwd(datt, soft_att, index_all, lrb, refb, finvq, finvk, Kk, side)
    K.scatter_ref_block(drefb, dref_out[L - 1], y1, x1, P, side)
    _, Cc, H, W = lq4.shape
    dlq4 = K.lr_blocks_bwd(dlrb, N, Cc, H, W, geo.py, geo.px, Kk, Kk)
    K.copy_rows(dlq4, Cc * H * W, dlq_out, Cc * H * W, N, Cc * H * W)


# ---------------------------------------------------------------------------
# whole network  NAFNetRefFusion.forward (:587-740)
# ---------------------------------------------------------------------------
def net_fwd(P, cfg, inp, ref):
    """inp, ref [N,3,H,W] -> (out [N,3,H,W], saved).  cfg: constructor kwargs."""
    n_enc = len(cfg['enc_blk_nums'])
    N = inp.shape[0]
    pyr, (H0, W0, Hp, Wp) = pyramids_fwd(P, cfg, inp, ref, n_enc, n_enc + 1)
    inp_p, geo = pyr.inp_p, pyr.geo
    # cat([x, warp], 1) of every fusion level (:719,727) without copies: the transfer kernel writes the warped reference
    # features into the second half of the level's concat buffer, the conv that produces x writes the first half
    chan = P['intro.weight'].shape[0]
    cats = []
    for lvl in range(n_enc + 1):
        Cl = pyr.ref_feats[lvl].shape[1]             # warped-feature channels (nf * 2^lvl) next to the chan * 2^lvl of x
        cats.append(torch.empty(N, (chan << lvl) + Cl, Hp >> lvl, Wp >> lvl, dtype=torch.float32, device=inp.device))
    warp, sv_masa = masa_fwd(pyr.lq_deep, pyr.ref_feats, N, geo, outs=[c[:, chan << lvl:] for lvl, c in enumerate(cats)])

    conv_fwd(inp_p, P['intro.weight'], P['intro.bias'], 1, 1, out=cats[0][:, :chan])
    sv_levels, skips = [], []
    for lvl in range(n_enc):
        x, sv_f = naf_seq_fwd(cats[lvl], P, f'masa_blk_enc.{lvl}.', cfg['reffusion_n_blocks'][lvl], c_out_last=chan)
        x, sv_e = naf_seq_fwd(x, P, f'encoders.{lvl}.', cfg['enc_blk_nums'][lvl])
        skips.append(x)
        conv_fwd(x, P[f'downs.{lvl}.weight'], P[f'downs.{lvl}.bias'], 2, 0, out=cats[lvl + 1][:, :2 * chan])
        sv_levels.append((sv_f, sv_e, x))
        chan *= 2
    cat = cats[n_enc]
    x, sv_fm = naf_seq_fwd(cat, P, 'masa_blk_middle.0.', cfg['reffusion_n_blocks'][n_enc], c_out_last=chan)
    x, sv_m = naf_seq_fwd(x, P, 'middle_blks.', cfg['middle_blk_num'])
    sv_dec = []
    for lvl in range(len(cfg['dec_blk_nums'])):
        xin = x
        x = up_fwd(xin, P[f'ups.{lvl}.0.weight'], skips[-1 - lvl])
        x, sv_d = naf_seq_fwd(x, P, f'decoders.{lvl}.', cfg['dec_blk_nums'][lvl])
        sv_dec.append((xin, sv_d))
    xe = x
    out_p = conv_fwd(xe, P['ending.weight'], P['ending.bias'], 1, 1, res=inp_p)
    out = out_p if (Hp, Wp) == (H0, W0) else K.pad_crop(out_p, H0, W0)
    saved = (N, (H0, W0, Hp, Wp), geo, pyr, None, None, sv_masa, warp, sv_levels, sv_fm, sv_m, sv_dec, xe)
    return out, saved


# ---------------------------------------------------------------------------- un-guided NAFNet (reference :305-386)
def tlsc_kernel_sizes(cfg, train_size):
    """pooling kernel of every U-Net level as `Local_Base.convert` fixes it (nafnet_local_arch.py:29-36,99-104, NAFNetLocal :756-768):
    the first forward runs on rand(train_size) with base_size = int(1.5 x train size), and each AvgPool2d keeps
    kernel = feature size at that forward * base_size // train size.  Level l sees the zero-padded train image >> l."""
    _, _, Ht, Wt = train_size
    n_enc = len(cfg['enc_blk_nums'])
    mult = 1 << n_enc
    Hp, Wp = -(-Ht // mult) * mult, -(-Wt // mult) * mult
    bh, bw = int(Ht * 1.5), int(Wt * 1.5)
    return [((Hp >> l) * bh // Ht, (Wp >> l) * bw // Wt) for l in range(n_enc + 1)]


def unet_fwd(P, cfg, inp, local=None):
    """`NAFNet.forward`: check_image_size (zero pad to a multiple of 2^len(encoders)) -> intro -> encoders / downs -> middle ->
    ups (+ skip) / decoders -> ending + inp -> crop.  Same block kernels as the guided network, no reference branch.
    local: per-level TLSC pooling kernels (tlsc_kernel_sizes) -- `NAFNetLocal`, inference only (saved state is empty)."""
    n_enc = len(cfg['enc_blk_nums'])
    N, _, H0, W0 = inp.shape
    mult = 1 << n_enc
    Hp, Wp = -(-H0 // mult) * mult, -(-W0 // mult) * mult
    inp_p = inp.contiguous() if (Hp, Wp) == (H0, W0) else K.pad_crop(inp.contiguous(), Hp, Wp)
    x = conv_fwd(inp_p, P['intro.weight'], P['intro.bias'], 1, 1)
    sv_levels, skips = [], []
    for lvl in range(n_enc):
        x, sv_e = naf_seq_fwd(x, P, f'encoders.{lvl}.', cfg['enc_blk_nums'][lvl], local=local[lvl] if local else None)
        skips.append(x)
        sv_levels.append((sv_e, x))
        x = conv_fwd(x, P[f'downs.{lvl}.weight'], P[f'downs.{lvl}.bias'], 2, 0)
    x, sv_m = naf_seq_fwd(x, P, 'middle_blks.', cfg['middle_blk_num'], local=local[n_enc] if local else None)
    sv_dec = []
    for lvl in range(len(cfg['dec_blk_nums'])):
        xin = x
        x = up_fwd(xin, P[f'ups.{lvl}.0.weight'], skips[-1 - lvl])
        x, sv_d = naf_seq_fwd(x, P, f'decoders.{lvl}.', cfg['dec_blk_nums'][lvl], local=local[n_enc - 1 - lvl] if local else None)
        sv_dec.append((xin, sv_d))
    out_p = conv_fwd(x, P['ending.weight'], P['ending.bias'], 1, 1, res=inp_p)
    out = out_p if (Hp, Wp) == (H0, W0) else K.pad_crop(out_p, H0, W0)
    return out, ((H0, W0, Hp, Wp), inp_p, sv_levels, sv_m, sv_dec, x)


def unet_bwd(dout, P, cfg, saved, G=None):
    """-> (dinp, G): gradient w.r.t. the input image (the `+ inp` skip and the intro conv) and every parameter"""
    with deferred_join():
        (H0, W0, Hp, Wp), inp_p, sv_levels, sv_m, sv_dec, xe = saved
        n_enc = len(cfg['enc_blk_nums'])
        G = {} if G is None else G
        dout = dout.contiguous()
        if (Hp, Wp) != (H0, W0):
            dout = K.pad_crop(dout, Hp, Wp)
        d, _, _ = conv_bwd(dout, xe, P['ending.weight'], 1, 1, into=(G, 'ending.weight', 'ending.bias'))
        dskips = [None] * n_enc
        for lvl in reversed(range(len(cfg['dec_blk_nums']))):
            xin, sv_d = sv_dec[lvl]
            d = naf_seq_bwd(d, P, f'decoders.{lvl}.', cfg['dec_blk_nums'][lvl], sv_d, G)
            dskips[n_enc - 1 - lvl] = d
            d, G[f'ups.{lvl}.0.weight'] = up_bwd(d, xin, P[f'ups.{lvl}.0.weight'])
        d = naf_seq_bwd(d, P, 'middle_blks.', cfg['middle_blk_num'], sv_m, G)
        for lvl in reversed(range(n_enc)):
            sv_e, x_skip = sv_levels[lvl]
            d, G[f'downs.{lvl}.weight'], G[f'downs.{lvl}.bias'] = conv_bwd(d, x_skip, P[f'downs.{lvl}.weight'], 2, 0,
                                                                          add_to_dx=dskips[lvl])
            d = naf_seq_bwd(d, P, f'encoders.{lvl}.', cfg['enc_blk_nums'][lvl], sv_e, G)
        dinp, G['intro.weight'], G['intro.bias'] = conv_bwd(d, inp_p, P['intro.weight'], 1, 1, need_dx=True, add_to_dx=dout)
        if (Hp, Wp) != (H0, W0):
            dinp = K.pad_crop(dinp, H0, W0)
        return dinp, G


def _pad_into(src, dst_view):
    """dst_view: dense [N,C,Hd,Wd] slice (contiguous along the batch)."""
    N, Cc, Hs, Ws = src.shape
    _, _, Hd, Wd = dst_view.shape
    if (Hs, Ws) == (Hd, Wd):
        K.copy_rows(src, Cc * Hs * Ws, dst_view, Cc * Hd * Wd, N, Cc * Hs * Ws)
    else:
        tmp = K.pad_crop(src, Hd, Wd)
        K.copy_rows(tmp, Cc * Hd * Wd, dst_view, Cc * Hd * Wd, N, Cc * Hd * Wd)


def net_bwd(dout, P, cfg, saved, G=None):
    """dout [N,3,H0,W0] -> dict of parameter gradients keyed like P.  `G` may be a caller's
    dict-like collector (e.g. parallel.GradSink, which starts the RCCL all-reduce of a
    gradient bucket as soon as its last tensor is stored)."""
    with deferred_join():
        return _net_bwd(dout, P, cfg, saved, G)


def _net_bwd(dout, P, cfg, saved, G):
    G = {} if G is None else G
    with late_leaves(G, level_ok=True):
        return _net_bwd_body(dout, P, cfg, saved, G)


def _net_bwd_body(dout, P, cfg, saved, G):
    N, (H0, W0, Hp, Wp), geo, pyr, _, _, sv_masa, warp, sv_levels, sv_fm, sv_m, sv_dec, xe = saved
    n_enc = len(cfg['enc_blk_nums'])
    inp_p = pyr.inp_p
    dout = dout.contiguous()
    if (Hp, Wp) != (H0, W0):
        dout = K.pad_crop(dout, Hp, Wp)
    # ending conv (+inp residual has no parameter gradient)
    d, G['ending.weight'], G['ending.bias'] = conv_bwd(dout, xe, P['ending.weight'], 1, 1)
    dskips = [None] * n_enc
    for lvl in reversed(range(len(cfg['dec_blk_nums']))):
        xin, sv_d = sv_dec[lvl]
        d = naf_seq_bwd(d, P, f'decoders.{lvl}.', cfg['dec_blk_nums'][lvl], sv_d, G)
        level_end(G)
        dskips[n_enc - 1 - lvl] = d                    # gradient of `x + enc_skip` w.r.t. the skip
        d, _ = up_bwd(d, xin, P[f'ups.{lvl}.0.weight'], into=(G, f'ups.{lvl}.0.weight'))
    d = naf_seq_bwd(d, P, 'middle_blks.', cfg['middle_blk_num'], sv_m, G)
    dcat = naf_seq_bwd(d, P, 'masa_blk_middle.0.', cfg['reffusion_n_blocks'][n_enc], sv_fm, G)
    level_end(G)
    dwarp = [None] * 5
    chan = dcat.shape[1] // 2
    dwarp[n_enc] = dcat[:, chan:]
    d = dcat[:, :chan]                                 # batch-strided view: every consumer takes an image stride
    for lvl in reversed(range(n_enc)):
        sv_f, sv_e, x_skip = sv_levels[lvl]
        # downs: gradient into the skip tensor, accumulated with the decoder-side skip gradient
        d, _, _ = conv_bwd(d, x_skip, P[f'downs.{lvl}.weight'], 2, 0, add_to_dx=dskips[lvl],
                           into=(G, f'downs.{lvl}.weight', f'downs.{lvl}.bias'))
        d = naf_seq_bwd(d, P, f'encoders.{lvl}.', cfg['enc_blk_nums'][lvl], sv_e, G)
        dcat = naf_seq_bwd(d, P, f'masa_blk_enc.{lvl}.', cfg['reffusion_n_blocks'][lvl], sv_f, G)
        level_end(G)
        chan = dcat.shape[1] // 2
        dwarp[lvl] = dcat[:, chan:]
        d = dcat[:, :chan]                             # batch-strided view: every consumer takes an image stride
    # intro conv: input image needs no gradient
    conv_bwd(d, inp_p, P['intro.weight'], 1, 1, need_dx=False, into=(G, 'intro.weight', 'intro.bias'))
    run_late_leaves(G, lambda: pyramids_bwd(dwarp, pyr, P, cfg, sv_masa, G))
    return G

#!/usr/bin/env python
"""Marginal WALL cost of a kernel family inside the captured step (timing only: the ablated step computes wrong numbers).
ABL = comma list of families whose launches are skipped (their outputs stay uninitialised workspace):
  leaf1x1   the deferred 1x1 weight gradients of the NAFBlocks (conv1 / conv4 / conv5 leaves)
  wg3x3     the plane weight gradients of the MASA-encoder ResidualBlocks
  conv3x3   the plane convolutions of the MASA-encoder ResidualBlocks (forward and data gradient)
  conv3x3bwd  only their data-gradient launches
  reduce    --
usage: ABL=leaf1x1 python profiles/probe_ablate.py   -> one line `ABL=<..> <ms/step>`"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from textualdegremoval_amd import kernels as K  # noqa: E402

abl = [s for s in os.environ.get('ABL', '').split(',') if s]

if 'leaf1x1' in abl:
    _cw = K.conv_wgrad

    def conv_wgrad(x, dout, Cout, Cin, KH, stride=1, pad=0, gate=False, per_image=False, want_db=False, fp16_range=False):
        if KH == 1 and not per_image:
            g = torch.empty(1, Cout, Cin, 1, 1, dtype=torch.float32, device=x.device)
            return (g, torch.empty(Cout, dtype=torch.float32, device=x.device)) if want_db else g
        return _cw(x, dout, Cout, Cin, KH, stride, pad, gate, per_image, want_db, fp16_range)
    K.conv_wgrad = conv_wgrad

if 'wg3x3' in abl:
    def wgrad3x3_p16(x16, d16, want_db=False):
        Cc = x16.C
        g = torch.empty(Cc, Cc, 3, 3, dtype=torch.float32, device=x16.buf.device)
        return (g, torch.empty(Cc, dtype=torch.float32, device=g.device)) if want_db else g
    K.wgrad3x3_p16 = wgrad3x3_p16

if 'conv3x3bwd' in abl:
    _c3 = K.conv3x3_p16

    def conv3x3_p16_b(x16, wp, Mpad, Cout, bias=None, res=None, mask=None, relu=False, want32=True, want16=False, out32=None):
        if not K.BACKWARD_PHASE:          # forward convolutions run; only the data-gradient convolutions are skipped
            return _c3(x16, wp, Mpad, Cout, bias=bias, res=res, mask=mask, relu=relu, want32=want32, want16=want16, out32=out32)
        dev = x16.buf.device
        o32 = (out32 if out32 is not None else torch.empty(x16.N, Cout, x16.H, x16.W, dtype=torch.float32, device=dev)) if want32 else None
        o16 = K.P16.empty(x16.N, Cout, x16.H, x16.W, dev, x16.fmt) if want16 else None
        return o32, o16
    K.conv3x3_p16 = conv3x3_p16_b

if 'conv3x3' in abl:
    def conv3x3_p16(x16, wp, Mpad, Cout, bias=None, res=None, mask=None, relu=False, want32=True, want16=False, out32=None):
        dev = x16.buf.device
        o32 = (out32 if out32 is not None else torch.empty(x16.N, Cout, x16.H, x16.W, dtype=torch.float32, device=dev)) if want32 else None
        o16 = K.P16.empty(x16.N, Cout, x16.H, x16.W, dev, x16.fmt) if want16 else None
        return o32, o16
    K.conv3x3_p16 = conv3x3_p16

if 'finish' in abl:
    # the small deferred finishers: LayerNorm-gradient partial reductions, depthwise parameter-gradient reductions, conv5 / gamma chain
    def _lnf(ws, nparts, Cc):
        def fin():
            gw = torch.empty(Cc, dtype=torch.float32, device=ws.device)
            return gw, torch.empty_like(gw)
        return fin
    K._ln_partials_finish = _lnf
    _lib0 = K._lib.load()

    class _LibSkip:
        def __getattr__(self, name):
            if name == 'tdr_dw_param_finish':
                return lambda *a: 0
            return getattr(_lib0, name)
    _skip = _LibSkip()
    _load0 = K._lib.load
    K._lib.load = lambda: _skip

    def scaled_conv_param_grads(G, S, w, b, gamma):
        Cout, Cin = G.shape[-2], G.shape[-1]
        return (torch.empty(Cout, Cin, dtype=torch.float32, device=G.device), torch.empty(Cout, dtype=torch.float32, device=G.device),
                torch.empty(Cout, dtype=torch.float32, device=G.device))
    K.scaled_conv_param_grads = scaled_conv_param_grads

import bench  # noqa: E402

sys.argv = [sys.argv[0], '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-f32-exact', '--no-roofline', '--no-matcher-active']
import io
import contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
line = [ln for ln in buf.getvalue().splitlines() if ln.startswith('{')][-1]
print(f"ABL={','.join(abl) or '-'} {json.loads(line)['ms_per_step']:.2f}")

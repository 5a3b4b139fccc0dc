"""Stage-A (image-to-text mapping) pieces timed on one MI355X:
CLIP ViT image encoder forward (ViT-H/14 1280/32L/16h and ViT-L/14 1024/24L/16h) on 4 x 3 x 224 x 224, Mapper(1280 -> 1024,
20 words) forward + backward, and the injected cross-attention forward + backward at the SD-2.1 UNet shapes
(context [4, 77, 1024]).  The UNet / VAE / text encoder themselves are third-party and absent, so this is
a per-piece measurement, not a train step.  Random-init weights; prints one JSON object."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textualdegremoval_amd import kernels as K  # noqa: E402
from textualdegremoval_amd.clip_vision import ClipVisionEncoder, random_clip_state_dict  # noqa: E402
from textualdegremoval_amd.i2t import Mapper, cross_attention  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    res = {'math': K.MATH}
    B = 4
    img = torch.rand(B, 3, 512, 512, device='cuda')
    for name, (hid, inter, layers, heads, act, flop) in {'vit_h14': (1280, 5120, 32, 16, 'gelu', 0.334e12),
                                                          'vit_l14': (1024, 4096, 24, 16, 'quick_gelu', None)}.items():
        enc = ClipVisionEncoder(random_clip_state_dict(hid, inter, layers), 'cuda', heads, act=act)
        ms = timeit(lambda: enc.encode(img))
        res[name + '_fwd_ms_bs4'] = ms
        if flop:
            res[name + '_fwd_tflops'] = flop * B / (ms * 1e-3) / 1e12
        try:      # the frozen encoder as one hipGraph (how a captured trainer step would hold it)
            from textualdegremoval_amd import kernels as KK
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                enc.encode(img)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g, refs = torch.cuda.CUDAGraph(), []
            with torch.cuda.graph(g, capture_error_mode='thread_local'), KK.workspace_capture(refs):
                held = enc.encode(img)
            torch.cuda.synchronize()
            gms = timeit(g.replay)
            res[name + '_fwd_ms_bs4_hipgraph'] = gms
            if flop:
                res[name + '_fwd_tflops_hipgraph'] = flop * B / (gms * 1e-3) / 1e12
            del g, held
        except Exception as e:  # noqa: BLE001
            res[name + '_fwd_ms_bs4_hipgraph'] = f'capture failed: {type(e).__name__}: {str(e)[:120]}'
        if name == 'vit_h14':
            tok = enc.encode(img)
        del enc
    mp = Mapper(1280, 1024, 20).cuda()

    def mapper_step():
        for p in mp.parameters():
            p.grad = None
        out = mp([tok])
        out.backward(torch.ones_like(out))
    res['mapper20_fwd_bwd_ms_bs4'] = timeit(mapper_step, n=3, warm=1)
    # the same forward + backward replayed as one hipGraph (what a trainer's captured step does: the ~1 200 launches and the four
    # stream lanes become graph nodes / branches, no host launch time)
    try:
        from textualdegremoval_amd import kernels as KK
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                mapper_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        refs = []
        with torch.cuda.graph(g, capture_error_mode='thread_local'), KK.workspace_capture(refs):
            mapper_step()
        torch.cuda.synchronize()
        res['mapper20_fwd_bwd_ms_bs4_hipgraph'] = timeit(g.replay, n=5, warm=2)
    except Exception as e:  # noqa: BLE001
        res['mapper20_fwd_bwd_ms_bs4_hipgraph'] = f'capture failed: {type(e).__name__}: {str(e)[:120]}'
    ctx = torch.randn(B, 77, 1024, device='cuda', requires_grad=True)
    for Tq, dim, heads in ((4096, 320, 5), (1024, 640, 10), (256, 1280, 20), (64, 1280, 20)):
        g = torch.Generator().manual_seed(dim)
        P = {'to_q.weight': torch.randn(dim, dim, generator=g) * dim ** -0.5,
             'to_k_global.weight': torch.randn(dim, 1024, generator=g) / 32, 'to_v_global.weight': torch.randn(dim, 1024, generator=g) / 32,
             'to_out.0.weight': torch.randn(dim, dim, generator=g) * dim ** -0.5, 'to_out.0.bias': torch.zeros(dim)}
        P = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
        hid = torch.randn(B, Tq, dim, device='cuda', requires_grad=True)

        def xstep():
            out = cross_attention(P, hid, ctx, heads, 64 ** -0.5)
            out.backward(torch.ones_like(out))
        res[f'xattn_T{Tq}_d{dim}_fwd_bwd_ms_bs4'] = timeit(xstep)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

"""How fast is the zero fill of the level-0 feature gradient (8 x 32 x 512 x 512 fp32 = 268 MB)?  torch's fill kernel inside a hipGraph
against hipMemsetAsync on the same stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]


def bench(name, f, n=20):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f'{name:50s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us', flush=True)


for shape in [(8, 32, 512, 512), (8, 64, 256, 256), (4, 32, 512, 512)]:
    x = torch.empty(*shape, device='cuda')
    mb = x.numel() * 4 / 1e6
    bench(f'{mb:.0f} MB zero_()', lambda: x.zero_())
    bench(f'{mb:.0f} MB zeros_like (fresh allocation)', lambda: torch.zeros_like(x))
    bench(f'{mb:.0f} MB hipMemsetAsync', lambda: hip.hipMemsetAsync(x.data_ptr(), 0, x.numel() * 4, torch.cuda.current_stream().cuda_stream))
    y = torch.zeros_like(x)
    bench(f'{mb:.0f} MB copy_ of zeros', lambda: x.copy_(y))

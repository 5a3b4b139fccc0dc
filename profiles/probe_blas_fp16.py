import torch
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
P = 31488
for name, N, K in (('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072), ('big', 8192, 8192)):
    x = torch.randn(P if name != 'big' else 8192, K, device='cuda', dtype=torch.float16)
    w = torch.randn(N, K, device='cuda', dtype=torch.float16)
    us = timeit(lambda: torch.matmul(x, w.t()))
    print(f'hipBLASLt/rocBLAS fp16 {name}: {us:7.1f} us {2 * x.shape[0] * N * K / us / 1e6:6.0f} TFLOP/s')

"""The product's dense GEMM on the shapes tools/bench/micro/gemm8p runs (plain C = A W^T, bf16, random uniform [-1, 1))."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/bench')
from emote_hack_amd import ops as o
dev, dt = 'cuda', torch.bfloat16
def timeit(f, n):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (98304, 2560, 320), (24576, 5120, 640), (6144, 10240, 1280), (98304, 1024, 320), (24576, 1280, 3200)):
    a = torch.rand(M, K, device=dev).mul_(2).sub_(1).to(dt); w = torch.rand(N, K, device=dev).mul_(2).sub_(1).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    us = timeit(lambda: o.gemm(a, w, None, out=out), 20 if K >= 4096 else 50)
    print(f"product M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us ({2.0 * M * N * K / us / 1e6:6.0f} TF/s)", flush=True)

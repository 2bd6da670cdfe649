"""cfg5 (768x768, 24 frames per window, 2 CFG rows = 48 images) 3x3 convs of the 24 x 24 level - frames that are not whole 8 x 16
patches: halo kernel with an overlapped last patch column (default) vs the im2col loader (split_k=2 forces the GEMM path)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev, dt = 'cuda', torch.float16


def run(n, H, W, Cin, N, up=False, **kw):
    xs = [torch.randn(n * H * W, Cin, device=dev, dtype=dt) for _ in range(4)]
    w = torch.randn(N, 9 * Cin, device=dev, dtype=dt) / 50
    b = torch.randn(N, device=dev)
    k = 4 if up else 1
    def f(i): return o.conv3x3(xs[i % 4], w, b, n, H, W, upsample2x=up, **kw)
    for i in range(3): f(i)
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(0)
        with torch.cuda.graph(g, stream=s):
            for i in range(12): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 12 * 1e3
    M = k * n * H * W
    return us, 2.0 * M * N * 9 * Cin / us / 1e6


for (n, H, W, Cin, N, up) in ((48, 24, 24, 1280, 1280, False), (48, 24, 24, 2560, 1280, False), (48, 24, 24, 1920, 1280, False), (48, 12, 12, 1280, 1280, True),
                              (48, 48, 48, 640, 640, False), (48, 12, 12, 1280, 1280, False)):
    a = run(n, H, W, Cin, N, up)
    b = run(n, H, W, Cin, N, up, split_k=2)
    print(f"conv{' up' if up else '   '} n={n} {H}x{W} Cin={Cin} N={N}: planned {a[0]:8.1f} us {a[1]:6.0f} TF/s | im2col (split_k 2) {b[0]:8.1f} us {b[1]:6.0f} TF/s", flush=True)

"""A/B of the halo conv's column split at N = 320 (16-row patches): 128 + 192 columns (two launches of full-width blocks, the product's plan)
against 128 + 128 + 64 (emo_gemm_params.tile bit 3 keeps the 64-column remainder launch).  bf16, HIP-graph timing, operands rotated."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev, dt = 'cuda', torch.bfloat16


def run(n, H, W, Cin, N, tile, up=False, res=True):
    xs = [torch.randn(n * H * W, Cin, device=dev, dtype=dt) for _ in range(4)]
    w = torch.randn(N, 9 * Cin, device=dev, dtype=dt) / 50
    b = torch.randn(N, device=dev)
    k = 4 if up else 1
    rs = [torch.randn(k * n * H * W, N, device=dev, dtype=dt) for _ in range(4)]
    def f(i): return o.conv3x3(xs[i % 4], w, b, n, H, W, residual=rs[i % 4] if res and not up else None, upsample2x=up, tile=tile)
    for i in range(3): f(i)
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(0)
        with torch.cuda.graph(g, stream=s):
            for i in range(12): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 12 * 1e3)
    M = k * n * H * W
    return best, 2.0 * M * N * 9 * Cin / best / 1e6


for (n, H, W, Cin, N, up) in ((24, 64, 64, 320, 320, False), (24, 64, 64, 640, 320, False), (24, 64, 64, 960, 320, False), (12, 64, 64, 320, 320, False),
                              (24, 32, 32, 320, 320, True), (24, 64, 64, 320, 192, False), (24, 32, 32, 640, 640, False)):
    a = run(n, H, W, Cin, N, 2 | 16, up)
    b = run(n, H, W, Cin, N, 2 | 8, up)
    print(f"conv{' up' if up else '   '} M={(4 if up else 1) * n * H * W:6d} N={N:4d} Cin={Cin:4d}: 128+192 {a[0]:7.1f} us {a[1]:5.0f} TF/s | 128+128+64 {b[0]:7.1f} us {b[1]:5.0f} TF/s | {100 * (b[0] / a[0] - 1):+5.1f} %", flush=True)

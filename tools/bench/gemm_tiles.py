#!/usr/bin/env python3
"""Tile-shape A/B of the dense GEMM and the halo conv on the UNet's own shapes (bf16, MI355X).

Every (shape, tile) pair is timed as a HIP graph of 12 launches rotating over 4 operand sets (so that A / C are not served
from the infinity cache), replayed twice; prints us per launch and TFLOP/s.  emo_gemm_params.tile pins the tile shape
(1 64x64, 2 128x128, 3 128x160, 4 256x256, 5 256x160, 6 256x320; conv: 1 = 8-row patches, 2 = 16-row patches)."""
import sys

import torch

sys.path.insert(0, ".")
from emote_hack_amd import ops as o  # noqa: E402

dev, dt = "cuda", torch.bfloat16
NROT, NL = 4, 12


def timeit(f):
    for i in range(2):
        f(i)
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(0)
        with torch.cuda.graph(g, stream=s):
            for i in range(NL):
                f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / NL * 1e3)
    return best


def dense(M, N, K, tiles, res=False, geglu=False, ln=False):
    a = [torch.randn(M, K, device=dev, dtype=dt) for _ in range(NROT)]
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    no = N // 2 if geglu else N
    r = [torch.randn(M, no, device=dev, dtype=dt) for _ in range(NROT)] if res else None
    out = [torch.empty(M, no, device=dev, dtype=dt) for _ in range(NROT)]
    cs = w.float().sum(1).contiguous()
    st = [o.layer_norm_stats(x_) for x_ in a] if ln else None
    row = f"M={M:6d} N={N:5d} K={K:5d} {'geglu' if geglu else '     '} {'res' if res else '   '} {'ln' if ln else '  '} |"
    for t in tiles:
        try:
            us = timeit(lambda i: o.gemm(a[i % NROT], w, b, geglu=geglu, residual=r[i % NROT] if res else None, out=out[i % NROT],
                                         tile=t, split_k=1, ln=(cs, st[i % NROT]) if ln else None))
            row += f" t{t}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF |"
        except Exception as ex:   # a tile the planner cannot serve
            row += f" t{t}: {type(ex).__name__} |"
    print(row, flush=True)


def conv(n, H, W, Cin, N):
    xs = [torch.randn(n * H * W, Cin, device=dev, dtype=dt) for _ in range(NROT)]
    w = (torch.randn(N, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    rs = [torch.randn(n * H * W, N, device=dev, dtype=dt) for _ in range(NROT)]
    M = n * H * W
    row = f"conv M={M:6d} N={N:5d} Cin={Cin:5d} |"
    for t in (1, 2):
        us = timeit(lambda i: o.conv3x3(xs[i % NROT], w, b, n, H, W, residual=rs[i % NROT], tile=t, split_k=1))
        row += f" ph{8 * t}: {us:7.1f}us {2.0 * M * N * 9 * Cin / us / 1e6:6.0f}TF |"
    print(row, flush=True)


def norms():
    for M, C in ((98304, 320), (24576, 640), (6144, 1280)):
        xs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(NROT)]
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        us_ln = timeit(lambda i: o.layer_norm(xs[i % NROT], g, b))
        us_st = timeit(lambda i: o.layer_norm_stats(xs[i % NROT]))
        us_gn5 = timeit(lambda i: o.group_norm(xs[i % NROT], g, b, 2, 32, 1e-5, True))
        us_gn4 = timeit(lambda i: o.group_norm(xs[i % NROT], g, b, 24, 32, 1e-6, False))
        mb = M * C * 2 / 1e6
        print(f"norms M={M:6d} C={C:5d} ({mb:5.1f} MB): layernorm {us_ln:6.1f}us ({2 * mb / us_ln:5.2f} TB/s r+w) | ln_stats {us_st:6.1f}us "
              f"({mb / us_st:5.2f} TB/s r) | GN 5-D+SiLU {us_gn5:6.1f}us ({3 * mb / us_gn5:5.2f} TB/s 2r+w) | GN per-frame {us_gn4:6.1f}us", flush=True)


if __name__ == "__main__":
    import os
    if os.environ.get("GEMM_TILES_QUICK"):   # the planned tile only, the shapes that dominate the step
        for args, kw in (((98304, 320, 320), dict(res=True)), ((98304, 960, 320), dict(ln=True)), ((98304, 320, 1280), dict(res=True)),
                         ((24576, 640, 640), dict(res=True)), ((24576, 640, 2560), dict(res=True)), ((6144, 1280, 1280), dict(res=True)),
                         ((6144, 1280, 5120), dict(res=True)), ((98304, 2560, 320), dict(geglu=True, ln=True)),
                         ((24576, 5120, 640), dict(geglu=True, ln=True)), ((6144, 10240, 1280), dict(geglu=True, ln=True))):
            dense(*args, (0,), **kw)
        raise SystemExit(0)
    norms()
    T = (0, 2, 3, 4)
    dense(98304, 320, 320, T, res=True)
    dense(98304, 960, 320, T)
    dense(98304, 960, 320, T, ln=True)
    dense(98304, 640, 320, T, ln=True)
    dense(98304, 320, 1280, T, res=True)
    dense(24576, 640, 640, T, res=True)
    dense(24576, 1920, 640, T, ln=True)
    dense(24576, 640, 2560, T, res=True)
    dense(6144, 1280, 1280, T, res=True)
    dense(6144, 3840, 1280, T, ln=True)
    dense(6144, 1280, 5120, T, res=True)
    G = (0, 4, 2)
    dense(98304, 2560, 320, G, geglu=True)
    dense(98304, 2560, 320, G, geglu=True, ln=True)
    dense(24576, 5120, 640, G, geglu=True)
    dense(24576, 5120, 640, G, geglu=True, ln=True)
    dense(6144, 10240, 1280, G, geglu=True, ln=True)
    dense(8192, 8192, 8192, (4, 6, 2))
    conv(24, 64, 64, 320, 320)
    conv(24, 64, 64, 640, 320)
    conv(24, 64, 64, 960, 320)
    conv(24, 32, 32, 640, 640)
    conv(24, 32, 32, 1280, 640)
    conv(24, 32, 32, 1920, 640)
    conv(24, 16, 16, 1280, 1280)
    conv(24, 16, 16, 2560, 1280)
    conv(10, 64, 64, 320, 320)

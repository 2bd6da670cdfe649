#!/usr/bin/env python3
"""Yard-stick, not product: the path's heaviest dense-GEMM and 3x3-conv shapes on the vendor libraries (hipBLASLt through
torch.nn.functional.linear, MIOpen through conv2d, both bf16 with bias) next to emo_gemm / emo conv3x3 on the same operands.
Same timing scheme as gemm_tiles.py (HIP graph of 12 launches rotating over 4 operand sets).  The libraries run the plain
op only (no GEGLU / LayerNorm fold / residual epilogue), so the fused shapes are compared on their GEMM part."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from emote_hack_amd import ops as o  # noqa: E402
from tools.bench.gemm_tiles import NROT, timeit  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def dense(M, N, K):
    a = [torch.randn(M, K, device=dev, dtype=dt) for _ in range(NROT)]
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b32 = torch.randn(N, device=dev)
    b16 = b32.to(dt)
    out = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NROT)]
    us_e = timeit(lambda i: o.gemm(a[i % NROT], w, b32, out=out[i % NROT]))
    us_l = timeit(lambda i: torch.addmm(b16, a[i % NROT], w.t(), out=out[i % NROT]))
    fl = 2.0 * M * N * K / 1e6
    print(f"dense M={M:6d} N={N:5d} K={K:5d} | emo {us_e:7.1f}us {fl / us_e:6.0f}TF | hipBLASLt {us_l:7.1f}us {fl / us_l:6.0f}TF | "
          f"emo/lib time {us_e / us_l:.2f}", flush=True)


def conv(n, H, W, Cin, N):
    M = n * H * W
    xs = [torch.randn(M, Cin, device=dev, dtype=dt) for _ in range(NROT)]
    w = (torch.randn(N, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
    b32 = torch.randn(N, device=dev)
    us_e = timeit(lambda i: o.conv3x3(xs[i % NROT], w, b32, n, H, W))
    # MIOpen: NHWC storage (channels_last), weight (N, Cin, 3, 3) channels_last
    xl = [x.view(n, H, W, Cin).permute(0, 3, 1, 2) for x in xs]
    wl = w.view(N, 3, 3, Cin).permute(0, 3, 1, 2)
    b16 = b32.to(dt)
    us_l = timeit(lambda i: F.conv2d(xl[i % NROT], wl, b16, padding=1))
    fl = 2.0 * M * N * 9 * Cin / 1e6
    print(f"conv  M={M:6d} N={N:5d} Cin={Cin:4d} | emo {us_e:7.1f}us {fl / us_e:6.0f}TF | MIOpen    {us_l:7.1f}us {fl / us_l:6.0f}TF | "
          f"emo/lib time {us_e / us_l:.2f}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    for M, N, K in ((98304, 320, 320), (98304, 960, 320), (98304, 2560, 320), (98304, 320, 1280), (24576, 640, 640),
                    (24576, 1920, 640), (24576, 5120, 640), (24576, 640, 2560), (6144, 1280, 1280), (6144, 3840, 1280),
                    (6144, 10240, 1280), (6144, 1280, 5120), (1536, 1280, 1280), (1536, 10240, 1280), (8192, 8192, 8192)):
        dense(M, N, K)
    if "--dense-only" in sys.argv:
        raise SystemExit(0)
    for n, H, W, Cin, N in ((24, 64, 64, 320, 320), (24, 64, 64, 640, 320), (24, 32, 32, 640, 640), (24, 32, 32, 1280, 640),
                            (24, 16, 16, 1280, 1280), (24, 16, 16, 2560, 1280), (24, 8, 8, 1280, 1280)):
        try:
            conv(n, H, W, Cin, N)
        except Exception as ex:
            print(f"conv {n}x{H}x{W} Cin={Cin} N={N}: {type(ex).__name__}: {ex}", flush=True)

#!/usr/bin/env python3
"""A/B harness for GEMM variant builds (tools/bench/build_variant.sh NAME flags; EMO_HIP_LIB=emote_hack_amd/lib/variants/NAME.so):
a correctness check of the tiles against an f32 reference, then us / TFLOP/s per (shape, tile) on the path's shapes.  Tile codes:
tile + 16 * gm (gm = rows per band of the tile order, 0 = planned, 1 = n fastest).  With
tools/bench/patches/gemm_staged_loader.patch applied (PATCH=... build_variant.sh staged) tiles 8 / 9 / 10 are the register-staged
256x256 / 128x128 / 128x160 variants."""
import sys

import torch

sys.path.insert(0, ".")
from emote_hack_amd import ops as o  # noqa: E402
from tools.bench.gemm_tiles import NROT, dense, timeit  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def check():
    torch.manual_seed(1)
    for M, N, K in ((1000, 320, 320), (4096, 640, 640), (777, 1280, 1280), (512, 2560, 320), (3000, 5120, 640)):
        a = torch.randn(M, K, device=dev, dtype=dt)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev, dtype=dt)
        want = a.float() @ w.float().t() + b + r.float()
        for t in (0, 1, 2, 3, 4):
            got = o.gemm(a, w, b, residual=r, tile=t, split_k=1).float()
            err = float((got - want).abs().max())
            print(f"check M={M} N={N} K={K} tile {t}: max err {err:.4f} {'OK' if err < 0.06 else 'FAIL'}", flush=True)


if __name__ == "__main__":
    check()
    T = (0, 2, 3, 4)   # tile + 16 * gm (gm: rows per band of the tile order; 0 = planned, 1 = n fastest)
    for args, kw in (((8192, 8192, 8192), {}), ((98304, 320, 320), dict(res=True)), ((98304, 960, 320), dict(ln=True)),
                     ((98304, 320, 1280), dict(res=True)), ((24576, 640, 640), dict(res=True)), ((24576, 1920, 640), dict(ln=True)),
                     ((24576, 640, 2560), dict(res=True)), ((6144, 1280, 1280), dict(res=True)), ((6144, 3840, 1280), dict(ln=True)),
                     ((6144, 1280, 5120), dict(res=True)), ((98304, 2560, 320), dict(geglu=True, ln=True)),
                     ((24576, 5120, 640), dict(geglu=True, ln=True)), ((6144, 10240, 1280), dict(geglu=True, ln=True)),
                     ((1536, 1280, 1280), dict(res=True)), ((1536, 10240, 1280), dict(geglu=True, ln=True))):
        dense(*args, T, **kw)

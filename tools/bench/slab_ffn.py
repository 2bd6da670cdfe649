"""Does running the GEGLU feed-forward slab by slab (the 4C-wide intermediate stays in the 256 MB Infinity Cache) beat one
full-M launch per GEMM?  ff1 (LN-folded GEGLU, N = 8C) -> ff2 (N = C, K = 4C, + residual) at the 64x64 / 32x32 / 16x16 levels."""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools/bench")
from emote_hack_amd import ops as o
from gemm_tiles import timeit
dev, dt = "cuda", torch.bfloat16
NROT = 3


def run(M, C, slabs):
    xs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(NROT)]
    w1 = (torch.randn(8 * C, C, device=dev) / C ** 0.5).to(dt); b1 = torch.randn(8 * C, device=dev); cs = w1.float().sum(1).contiguous()
    w2 = (torch.randn(C, 4 * C, device=dev) / (4 * C) ** 0.5).to(dt); b2 = torch.randn(C, device=dev)
    g = [torch.empty(M, 4 * C, device=dev, dtype=dt) for _ in range(NROT)]
    out = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(NROT)]
    st = [o.layer_norm_stats(x) for x in xs]
    row = f"M={M} C={C}:"
    for ns in slabs:
        R = M // ns
        def f(i):
            j = i % NROT
            for s in range(ns):
                sl = slice(s * R, (s + 1) * R)
                o.gemm(xs[j][sl], w1, b1, geglu=True, ln=(cs, st[j][sl]), out=g[j][sl])
                o.gemm(g[j][sl], w2, b2, residual=xs[j][sl], out=out[j][sl])
        us = timeit(f)
        row += f"  {ns} slab(s): {us:7.1f} us"
    print(row, flush=True)


run(98304, 320, (1, 2, 4, 8, 12)); run(24576, 640, (1, 2, 4)); run(6144, 1280, (1, 2))

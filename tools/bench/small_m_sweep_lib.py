"""Tile x split-K sweep of the under-filled GEMMs (the 8x8 / 16x16 levels: M = 1536 / 6144) - these are latency-bound, not
MFMA-bound; run once per ring-depth variant (EMO_HIP_LIB=emote_hack_amd/lib/variants/ns3.so ...)."""
import os
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from emote_hack_amd import ops as o  # noqa: E402
from gemm_tiles import timeit, NROT  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def dense(M, N, K, combos, res=False, geglu=False, ln=False):
    a = [torch.randn(M, K, device=dev, dtype=dt) for _ in range(NROT)]
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    no = N // 2 if geglu else N
    r = [torch.randn(M, no, device=dev, dtype=dt) for _ in range(NROT)] if res else None
    out = [torch.empty(M, no, device=dev, dtype=dt) for _ in range(NROT)]
    cs = w.float().sum(1).contiguous()
    st = [o.layer_norm_stats(x_) for x_ in a] if ln else None
    row = f"M={M:6d} N={N:5d} K={K:5d} {'geglu' if geglu else '     '} {'res' if res else '   '} {'ln' if ln else '  '} |"
    for t, s in combos:
        try:
            us = timeit(lambda i: o.gemm(a[i % NROT], w, b, geglu=geglu, residual=r[i % NROT] if res else None, out=out[i % NROT],
                                         tile=t, split_k=s, ln=(cs, st[i % NROT]) if ln else None))
            row += f" t{t}s{s}:{us:6.1f} |"
        except Exception as ex:
            row += f" t{t}s{s}: {type(ex).__name__} |"
    print(row, flush=True)


def conv(n, H, W, Cin, N, sks):
    xs = [torch.randn(n * H * W, Cin, device=dev, dtype=dt) for _ in range(NROT)]
    w = (torch.randn(N, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    rs = [torch.randn(n * H * W, N, device=dev, dtype=dt) for _ in range(NROT)]
    M = n * H * W
    row = f"conv M={M:6d} N={N:5d} Cin={Cin:5d} |"
    for t, s in sks:
        try:
            us = timeit(lambda i: o.conv3x3(xs[i % NROT], w, b, n, H, W, residual=rs[i % NROT], tile=t, split_k=s))
            row += f" t{t}s{s}:{us:6.1f} |"
        except Exception as ex:
            row += f" t{t}s{s}: {type(ex).__name__} |"
    print(row, flush=True)



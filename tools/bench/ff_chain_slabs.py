"""The product's feed-forward chain (LayerNorm-folded GEGLU writing g into the left 4C columns of the (rows, 5C) buffer [g | h], then the fused
ff.net.2 + proj_out GEMM over K = 5C with the block's residual) run over the whole M or slab by slab, so that a slab's [g | h] (5C wide) stays in
the 256 MB Infinity Cache between its producer and its consumer.  Run once per library: the product (non-temporal GEGLU stores) and the variant
built from tools/bench/patches/gemm_geglu_store_policy.patch with -DEMO_GEGLU_STORE_AUX=0 (default-policy GEGLU stores)."""
import os, sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools/bench")
from emote_hack_amd import ops as o
from gemm_tiles import timeit
dev, dt = "cuda", torch.bfloat16
NROT = 3


def run(M, C, slabs):
    hs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(NROT)]          # the residual stream h (attn2 output) - also the GEGLU's input
    xs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(NROT)]          # the transformer's input (final residual)
    w1 = (torch.randn(8 * C, C, device=dev) / C ** 0.5).to(dt); b1 = torch.randn(8 * C, device=dev); cs = w1.float().sum(1).contiguous()
    wt = (torch.randn(C, 5 * C, device=dev) / (5 * C) ** 0.5).to(dt); bt = torch.randn(C, device=dev)
    gh = [torch.empty(M, 5 * C, device=dev, dtype=dt) for _ in range(NROT)]
    for j in range(NROT):
        gh[j][:, 4 * C:].copy_(hs[j])
    out = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(NROT)]
    row = f"M={M:6d} C={C:5d}:"
    for ns in slabs:
        R = M // ns
        def f(i):
            j = i % NROT
            for s in range(ns):
                sl = slice(s * R, (s + 1) * R)
                h = gh[j][sl, 4 * C:]
                st = o.layer_norm_stats(h)
                o.gemm(h, w1, b1, geglu=True, ln=(cs, st), out=gh[j][sl, :4 * C])
                o.gemm(gh[j][sl], wt, bt, residual=xs[j][sl], out=out[j][sl])
        us = timeit(f)
        row += f"  {ns:2d} slab(s) {us:7.1f} us |"
    print(row, flush=True)


print("library:", os.environ.get("EMO_HIP_LIB", "product"), flush=True)
run(98304, 320, (1, 2, 3, 4, 6, 12)); run(24576, 640, (1, 2, 4)); run(6144, 1280, (1, 2))

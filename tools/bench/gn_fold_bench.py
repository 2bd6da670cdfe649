"""GroupNorm -> proj_in: the unfused pair (stats + apply, GEMM) against the fold (stats + per-instance weights, GEMM over the
raw rows), per level; us per call (HIP graph of 12 calls over 4 operand sets)."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from emote_hack_amd import ops as o  # noqa: E402
from gemm_tiles import timeit, NROT  # noqa: E402
dev, dt = "cuda", torch.bfloat16
for n, S, C in ((24, 4096, 320), (24, 1024, 640), (24, 256, 1280)):
    xs = [torch.randn(n * S, C, device=dev, dtype=dt) for _ in range(NROT)]
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    w = (torch.randn(C, C, device=dev) / C ** 0.5).to(dt)
    bias = torch.randn(C, device=dev)
    wn, rb = o.group_norm_fold_linear(xs[0], g, b, n, 32, 1e-6, w, bias)
    t = {}
    t["gn(stats+apply)"] = timeit(lambda i: o.group_norm(xs[i % NROT], g, b, n, 32, 1e-6, False))
    t["gemm plain"] = timeit(lambda i: o.gemm(xs[i % NROT], w, bias))
    t["pair"] = timeit(lambda i: o.gemm(o.group_norm(xs[i % NROT], g, b, n, 32, 1e-6, False), w, bias))
    t["stats+fold"] = timeit(lambda i: o.group_norm_fold_linear(xs[i % NROT], g, b, n, 32, 1e-6, w, bias))
    t["gemm slab+bias"] = timeit(lambda i: o.gemm(xs[i % NROT], wn, rb, w_slab_rows=S))
    t["gemm slab"] = timeit(lambda i: o.gemm(xs[i % NROT], wn, None, w_slab_rows=S))
    t["gemm rowbias"] = timeit(lambda i: o.gemm(xs[i % NROT], w, None, rowbias=rb, rows_per_batch=S))

    def folded(i):
        wn_, rb_ = o.group_norm_fold_linear(xs[i % NROT], g, b, n, 32, 1e-6, w, bias)
        return o.gemm(xs[i % NROT], wn_, rb_, w_slab_rows=S)
    t["folded"] = timeit(folded)
    print(f"n={n} S={S} C={C} | " + " | ".join(f"{k}: {v:6.1f}" for k, v in t.items()), flush=True)

"""GroupNorm (statistics + apply launches) with operands rotated out of the Infinity Cache: run once per library (EMO_HIP_LIB) - the product vs
a build whose apply-pass prologue requests its chunk partials 16 at a time instead of one per loop iteration."""
import os, sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools/bench')
from emote_hack_amd import ops as o
from gemm_tiles import timeit
dev, dt = 'cuda', torch.bfloat16
print('library:', os.environ.get('EMO_HIP_LIB', 'product'), flush=True)
for M, C, inst, silu in ((98304, 320, 2, True), (98304, 640, 2, True), (98304, 960, 2, True), (24576, 640, 2, True), (24576, 1280, 2, True), (6144, 1280, 2, True),
                         (98304, 320, 24, False), (24576, 640, 24, False)):
    xs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(4)]
    ys = [torch.empty_like(x) for x in xs]
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    us = timeit(lambda i: o.group_norm(xs[i % 4], g, b, inst, 32, 1e-5, silu, out=ys[i % 4]))
    print(f"GN M={M:6d} C={C:5d} inst={inst:3d}: {us:7.1f} us  {M * C * 2 * 3 / us / 1e6:5.2f} TB/s over its three passes", flush=True)
